"""Multi-GPU sharding of the horizon / shadow path (one process per GPU).

Every inner-domain cell is independent (horizon_comp.cpp:744-796), so the path
shards by contiguous row slabs with NO collective during traversal.  The only
exchange steps are
  1. one broadcast of the scene blob (vertices + LBVH, one contiguous HBM
     allocation) from the rank that built it -- ``torch.distributed`` backend
     "nccl" is RCCL over xGMI on ROCm;
  2. a final gather of per-rank result slabs.
The same functions run on the ``gloo`` backend with CPU tensors (tests).
"""
import numpy as np


def row_slabs(mask_or_rows, world_size):
    """Split rows into ``world_size`` contiguous slabs balanced by the number of cells
    to compute (``mask == 1`` per row).  Accepts a 2-D mask or a row count.
    Returns [(begin, end)] * world_size; slabs may be empty when rows < world_size."""
    if np.ndim(mask_or_rows) == 0:
        w = np.ones(int(mask_or_rows), np.int64)
    else:
        w = (np.asarray(mask_or_rows) == 1).sum(axis=1).astype(np.int64)
    n = w.shape[0]
    cum = np.concatenate([[0], np.cumsum(w)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        b = int(np.searchsorted(cum, target, side="left"))
        bounds.append(min(max(b, bounds[-1]), n))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def broadcast_scene(scene, device, src=0, group=None):
    """Broadcast a scene blob from rank ``src``; returns a Scene valid on this rank.

    ``scene`` is the built ``horayzon_amd.Scene`` on ``src`` and ``None`` elsewhere.
    The blob is received into a torch uint8 CUDA tensor that the adopted scene keeps
    alive."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from . import _lib
    rank = dist.get_rank(group)
    meta = torch.zeros(1, dtype=torch.int64, device="cuda:%d" % device)
    if rank == src:
        p, n = scene.blob()
        meta[0] = n
    dist.broadcast(meta, src=src, group=group)
    n = int(meta.item())
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:%d" % device)
    if rank == src:
        hiprt = C.CDLL("libamdhip64.so")   # already mapped; device-to-device copy into the send buffer
        rc = hiprt.hipMemcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(p), C.c_size_t(n), 3)
        if rc != 0:
            raise _lib.HorayzonHipError("hipMemcpy of the scene blob failed (%d)" % rc)
    dist.broadcast(buf, src=src, group=group)
    if rank == src:
        return scene
    return _lib.Scene.adopt(buf.data_ptr(), n, device, keepalive=buf)


def gather_rows(local, slabs, dst=0, group=None):
    """Gather per-rank row slabs (torch tensors, leading axis = rows of the slab) into
    the full array on ``dst``.  Works for CUDA tensors (RCCL) and CPU tensors (gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    max_rows = max(e - b for b, e in slabs)
    tail = tuple(local.shape[1:])
    padded = torch.zeros((max_rows,) + tail, dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    out = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    if dist.get_backend(group) == "nccl":
        # RCCL: gather is implemented through all_gather for portability across versions
        out = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(out, padded, group=group)
    else:
        dist.gather(padded, out, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([out[r][:slabs[r][1] - slabs[r][0]] for r in range(world)], dim=0)
