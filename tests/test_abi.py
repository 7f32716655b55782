"""CPU: the C-ABI library loads and exports exactly what include/pvamd.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess

import pytest

import pytorch_volumetric_amd as pv
from pytorch_volumetric_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pvamd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pvamd_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.load()
    assert lib.pvamd_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in lib.pvamd_build_info()


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 12
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pvamd.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and header disagree"


def test_no_library_kernels_inside_the_shared_object():
    """the spatial sort called rocprim::radix_sort_pairs from 1.5 M points on; the hot path is hand-written
    throughout now -- no rocPRIM / hipCUB / Thrust instantiation is linked into libpvamd.so."""
    out = subprocess.run(["nm", "-C", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for lib_name in ("rocprim", "hipcub", "thrust", "rocblas", "hipblas"):
        assert lib_name not in out.lower(), f"{lib_name} symbols in libpvamd.so"
    src = open(os.path.join(ROOT, "pytorch_volumetric_amd", "csrc", "sort.hip")).read()
    assert "#include <rocprim" not in src


def test_struct_layouts_match_the_header():
    """Compile a tiny C program against the header and compare sizeof/offsetof with the ctypes mirrors."""
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "pvamd.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(pvamd_grid_t), offsetof(pvamd_grid_t, dres), offsetof(pvamd_grid_t, fmin),
             offsetof(pvamd_grid_t, bb_min), offsetof(pvamd_grid_t, shape), offsetof(pvamd_grid_t, index_f64),
             offsetof(pvamd_grid_t, oob_mode));
      printf("%zu %zu %zu\n", sizeof(pvamd_mesh_t), offsetof(pvamd_mesh_t, F), offsetof(pvamd_mesh_t, ray_dir));
      return 0; }'''
    exe = "/tmp/_pvamd_layout"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    out = subprocess.run([exe], capture_output=True, check=True).stdout.decode().split()
    G, M = _lib.GridDesc, _lib.MeshDesc
    assert [int(x) for x in out[:7]] == [ctypes.sizeof(G), G.dres.offset, G.fmin.offset, G.bb_min.offset, G.shape.offset,
                                         G.index_f64.offset, G.oob_mode.offset]
    assert [int(x) for x in out[7:]] == [ctypes.sizeof(M), M.F.offset, M.ray_dir.offset]


def test_oracle_grid_layout_is_one_definition():
    """The checker's grid description lives in oracle/pvamd_oracle.h; the plain-C host test (tests/cabi/cabi_check.c) includes
    it and the ctypes mirror in oracle/oracle.py must agree with it (an outdated private copy in the C test once made the
    oracle read every grid after the first at the wrong stride)."""
    from oracle import oracle as orc
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "pvamd_oracle.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu\n", sizeof(oracle_grid_t), offsetof(oracle_grid_t, dres), offsetof(oracle_grid_t, fres),
             offsetof(oracle_grid_t, shape), offsetof(oracle_grid_t, rule), offsetof(oracle_grid_t, dbb_max));
      return 0; }'''
    exe = "/tmp/_pvamd_oracle_layout"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "oracle"), "-o", exe], input=src.encode(), check=True)
    out = [int(x) for x in subprocess.run([exe], capture_output=True, check=True).stdout.decode().split()]
    G = orc.OracleGrid
    assert out == [ctypes.sizeof(G), G.dres.offset, G.fres.offset, G.shape.offset, G.rule.offset, G.dbb_max.offset]
    assert "pvamd_oracle.h" in open(os.path.join(ROOT, "tests", "cabi", "cabi_check.c")).read()


def test_buffer_size_macros_match_the_python_mirrors():
    """The sizes the binding allocates are the header's macros (records in whole tiles, tile / group spheres, the scratch of
    the mesh entry points, the Morton-order scratch)."""
    src = r'''
    #include <stdio.h>
    #include "pvamd.h"
    int main(void) {
      long long v[] = {0, 1, 255, 256, 257, 15728, 99500, 1 << 20, 1 << 21, (1 << 24) + 5, 67108877};
      for (int i = 0; i < 11; ++i)
        printf("%lld %lld %lld %lld\n", (long long)PVAMD_REC_FLOATS(v[i]), (long long)PVAMD_TILES_FLOATS(v[i]),
               (long long)PVAMD_MESH_SCRATCH_BYTES(v[i]), (long long)PVAMD_MORTON_ORDER_SCRATCH_BYTES(v[i]));
      printf("%d %d %d %d\n", PVAMD_TRI_REC, PVAMD_TRI_TILE, PVAMD_TRI_GROUP, PVAMD_MESH_SCRATCH_GROUPS);
      return 0; }'''
    exe = "/tmp/_pvamd_sizes"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    out = [[int(x) for x in line.split()] for line in subprocess.run([exe], capture_output=True, check=True).stdout.decode().splitlines()]
    for n, row in zip((0, 1, 255, 256, 257, 15728, 99500, 1 << 20, 1 << 21, (1 << 24) + 5, 67108877), out):
        assert row == [_lib.rec_floats(n), _lib.tiles_floats(n), _lib.mesh_scratch_bytes(n),
                       4 * _lib.morton_order_scratch_words(n)], (n, row)
    assert out[-1] == [_lib.TRI_REC, _lib.TRI_TILE, _lib.TRI_GROUP, _lib.MESH_SCRATCH_GROUPS]


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.PvamdError):
        _lib.require_gpu()
    obj = pv.MeshObjectFactory(mesh=pv.mesh_io.box_mesh()) if hasattr(pv, "mesh_io") else None
    from pytorch_volumetric_amd import mesh_io
    obj = pv.MeshObjectFactory(mesh=mesh_io.box_mesh())
    with pytest.raises(_lib.PvamdError):
        pv.MeshSDF(obj)(torch.zeros(4, 3))


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under pytorch_volumetric_amd/ may import, load or link it."""
    pkg = os.path.join(ROOT, "pytorch_volumetric_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "libpvamd_oracle" not in text, f
                if f.endswith((".hip", ".h")) and f != "xform.hip":
                    assert '#include "../../oracle' not in text and "oracle/" not in text.replace("oracle/pvamd_oracle.c", ""), f
    deps = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True).stdout.decode()
    assert "oracle" not in deps


def test_no_wrong_result_experiments_in_the_product_sources_and_variants_are_refused():
    """timing experiments that return WRONG results are patches under tools/patches now, not #ifdefs in
    csrc/; an A/B build (tools/build_variant.sh) names itself through `pvamd_variant` and _lib.load() refuses it unless
    PVAMD_ALLOW_VARIANT=1.  The product library exports no such symbol."""
    import glob
    csrc = os.path.join(ROOT, "pytorch_volumetric_amd", "csrc")
    for path in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        text = open(path).read()
        for word in ("ABLATE", "ONLY_SEED", "WRONG results"):
            assert word not in text, f"{word} in {os.path.basename(path)}"
    out = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "pvamd_variant" not in out and "pvamd_debug" not in out
    assert "pvamd_variant" in open(os.path.join(csrc, "common.h")).read()
    assert "PVAMD_ALLOW_VARIANT" in open(os.path.join(ROOT, "pytorch_volumetric_amd", "_lib.py")).read()
