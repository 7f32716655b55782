"""pv.warm_up(): absorb the host stalls of a fresh process before a latency-sensitive loop starts (profiles/r04_stall.txt).

1. The sporadic 20-40 ms step that rounds 2 and 3 located but could not name is CPython's GENERATION-2 GARBAGE COLLECTION:
   `import torch` leaves ~170,000 collector-tracked objects behind, a full pass over them takes 30-40 ms on the GPU box's
   host, and it is triggered by allocation counts -- so it lands on some innocent call a few hundred calls into a loop
   (always the same one: call 681 of tools/stall_trace.py), inside whatever that call happens to allocate.  gc.freeze()
   moves everything alive now into the permanent generation; later passes look at what was allocated since (microseconds).
2. HIP loads device code lazily: the FIRST launch out of a code object that is not yet on the GPU runs
   hsa_executable_load_agent_code_object + hsa_executable_freeze + a walk over its symbols inside that hipLaunchKernel --
   27-83 ms for torch's own objects, a few ms for the ten of libpvamd.so (one per translation unit).  This function
   launches one tiny piece of work out of every translation unit so that those loads happen here.  (HIP's own switch,
   HIP_ENABLE_DEFERRED_LOADING=0, crashes this torch build at start-up.)
Separately, a container with a CPU quota below its visible CPU count (the GPU box: 256 CPUs visible, quota 16) throttles the
whole process when torch's 128 OpenMP threads spin after a parallel region: OMP_NUM_THREADS or OMP_WAIT_POLICY=passive.
"""
import gc

import numpy as np
import torch


def warm_up(device=None, freeze_gc=True):
    """Touch every kernel family once on tiny inputs (~0.3 s the first time, a few ms after), then -- unless freeze_gc is
    False -- gc.collect() + gc.freeze(): objects alive at this point are never scanned again by the cyclic collector (they
    are still freed by reference counting; only reference CYCLES among them would stay).  Returns the device used."""
    import pytorch_volumetric_amd as pv
    from pytorch_volumetric_amd import _lib, mesh_io
    from pytorch_volumetric_amd import transforms as tf
    lib = _lib.load()
    dev = _lib.require_gpu() if device is None else torch.device(device)
    with _lib.on_device(dev):
        obj = pv.MeshObjectFactory(mesh=mesh_io.uv_sphere_mesh(0.1, 12, 6))
        mesh = pv.MeshSDF(obj)
        g = torch.Generator().manual_seed(0)
        few = (torch.rand(64, 3, generator=g) * 0.5 - 0.25).to(dev)
        many = (torch.rand(40_000, 3, generator=g) * 0.5 - 0.25).to(dev)
        mesh(few)                                             # mesh.hip (few-points launches), sort.hip (one-block order)
        mesh(many)                                            # mesh.hip (main launch), sort.hip (counting sort)
        cached = pv.CachedSDF("warm_up", 0.02, obj.bounding_box(padding=0.06), mesh, device=dev, cache_path=None)  # cache build
        cached(few); cached(many); cached.outside_surface(few); cached(few.double())                                 # cached.hip
        rot = tf.Transform3d(matrix=torch.eye(4).repeat(6, 1, 1))
        comp = pv.ComposedSDF([cached, cached], rot)
        comp.set_transforms(rot, batch_dim=(3,))
        comp(few)                                             # composed.hip: per-lane kernel
        grids = comp._leaf_grids(dev)
        val = torch.empty((3, many.shape[0]), dtype=torch.float32, device=dev)
        grad = torch.empty((3, many.shape[0], 3), dtype=torch.float32, device=dev)
        for flags in (4, 4 | 16, 4 | 1):                      # the wave-tile kernel's three leaf loops
            _lib.check(lib.pvamd_composed_query(_lib.ptr(grids), 2, _lib.ptr(comp._tf_device(dev)), 3, _lib.ptr(many),
                                                many.shape[0], _lib.ptr(val), _lib.ptr(grad), None, flags, _lib.stream_ptr()),
                       "pvamd_composed_query")
        comp.bucket_points = True
        comp(many)                                            # packed records + un-permute, spatial order
        comp(few.double())                                    # float64 entry
        W = torch.eye(4, device=dev).unsqueeze(0)
        pv.batch_chamfer_dist(W, many[:4096], obj_sdf=cached)        # chamfer_grid.hip
        pv.batch_chamfer_dist(W, many[:4096], obj_factory=obj)       # mesh.hip chamfer kernels
        pv.sample_mesh_points(obj, num_points=32, seed=0, dbpath=None, device=dev)   # sample.hip
        vg = pv.VoxelGrid(0.05, [(-0.3, 0.3)] * 3, device=dev)       # voxelgrid.hip
        vg[few] = torch.ones(few.shape[0], device=dev)
        vg[few]
        # fk.hip + xform.hip: a two-frame chain through the one-launch configure and the round-3 pair of kernels
        joint = _lib.JointDesc()
        joint.parent, joint.jtype, joint.jcol, joint.leaf_slot = -1, 1, 0, 0
        joint.axis[2] = 1.0
        for k in (0, 5, 10):
            joint.origin[k] = 1.0
        table = torch.frombuffer(bytearray(bytes(joint)), dtype=torch.uint8).to(dev)
        q = torch.zeros((4, 1), dtype=torch.float32, device=dev)
        off = torch.eye(4, device=dev).unsqueeze(0).contiguous()
        scratch = torch.empty((1, 12, 4), dtype=torch.float32, device=dev)
        lw = torch.empty((4, 4, 4), dtype=torch.float32, device=dev)
        stack = torch.empty((4, 4, 4), dtype=torch.float32, device=dev)
        _lib.check(lib.pvamd_configure_chain(_lib.ptr(table), 1, _lib.ptr(q), 4, 1, _lib.ptr(off), 1, None, _lib.ptr(scratch),
                                             _lib.ptr(lw), _lib.ptr(stack), _lib.stream_ptr()), "pvamd_configure_chain")
        _lib.check(lib.pvamd_transform_stack(_lib.ptr(off), _lib.ptr(lw), 1, 4, _lib.ptr(stack), _lib.stream_ptr()),
                   "pvamd_transform_stack")
        torch.cuda.synchronize(dev)
    if freeze_gc:
        gc.collect()
        gc.freeze()
    return dev
