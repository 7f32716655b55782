#!/bin/bash
cd /root/repo
python tools/sort_probe.py 2>&1 | grep "torch.sort" | tail -4
tools/trace_c5.sh r4trace_c5d 2>&1 | tail -14 | head -12
