#!/usr/bin/env python
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide prescribes) into the
per-launch HBM-side traffic of one kernel.  gfx950 correction: FETCH_SIZE counts 128-B requests at 64 B, i.e. reports
half of a wide coalesced stream (MI355X_MICROARCH.md, HBM section) -- calibrated here on the 64M-point launch, whose
805.3 MB point stream reads back as 405.8 MB raw (factor 1.98) -- so fetch bytes = 2 x FETCH_SIZE x 1024.
WRITE_SIZE x 1024 matches the 16 B/point written exactly and needs no correction.
Usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> <points> > out.json
"""
import csv
import json
import sys


def values(path, kernel, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            out.append(float(r["Counter_Value"]))
    return out


def main():
    fetch_csv, write_csv, kernel, points = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    f, w = values(fetch_csv, kernel, "FETCH_SIZE"), values(write_csv, kernel, "WRITE_SIZE")
    f_mean, w_mean = sum(f) / len(f), sum(w) / len(w)
    fetch_bytes, write_bytes = 2.0 * f_mean * 1024.0, w_mean * 1024.0
    print(json.dumps({
        "kernel": kernel, "points": points, "launches_sampled": [len(f), len(w)],
        "FETCH_SIZE_KB_raw_mean": f_mean, "WRITE_SIZE_KB_mean": w_mean,
        "fetch_bytes_per_launch_corrected_x2": fetch_bytes, "write_bytes_per_launch": write_bytes,
        "hbm_bytes_per_launch": fetch_bytes + write_bytes,
        "algorithmic_bytes_per_launch": 28 * points,
        "note": "fetch includes the 781 KB voxel grid once per XCD L2 (8 x 0.78 MB) on every launch"}, indent=1))


if __name__ == "__main__":
    main()
