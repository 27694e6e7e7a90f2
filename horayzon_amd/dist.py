"""Multi-GPU sharding of the horizon / shadow path (one process per GPU).

Every inner-domain cell is independent (horizon_comp.cpp:744-796), so the path
shards by contiguous row slabs with NO collective during traversal.  The only
exchange steps are
  1. one broadcast of the scene blob (vertices + LBVH, one contiguous HBM
     allocation) from the rank that built it -- ``torch.distributed`` backend
     "nccl" is RCCL over xGMI on ROCm;
  2. a final gather of per-rank result slabs (small outputs: SVF, 4 B / cell; a
     horizon array is written by each rank into its own slice and never gathered).
``sharded_rows`` is the per-rank body of such a job; ``bench.py --workload c5`` and the
world_size-2 ``gloo`` test (tests/test_dist_gloo.py, CPU tensors) both run it.
"""
import time

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


def broadcast_blob(buf, nbytes_if_src, device, src=0, group=None):
    """Broadcast one contiguous byte buffer (the scene blob) from rank ``src``.

    ``buf`` is a uint8 torch tensor on ``src`` (any device) and ignored elsewhere;
    ``device`` is the torch device the receivers allocate on ("cuda:k" for RCCL, "cpu"
    for gloo).  Two collectives: the size, then the bytes.  Returns the tensor that
    holds the blob on this rank (``buf`` itself on ``src``)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    meta = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        meta[0] = int(nbytes_if_src)
    dist.broadcast(meta, src=src, group=group)
    n = int(meta.item())
    if rank != src:
        buf = torch.empty(n, dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src, group=group)
    return buf


def _hip_blob_tensor(scene, device):
    """The scene's blob copied into a torch uint8 CUDA tensor (send buffer of the broadcast)."""
    import ctypes as C
    import torch
    from . import _lib
    p, n = scene.blob()
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:%d" % device)
    hiprt = C.CDLL("libamdhip64.so")   # already mapped; device-to-device copy
    rc = hiprt.hipMemcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(p), C.c_size_t(n), 3)
    if rc != 0:
        raise _lib.HorayzonHipError("hipMemcpy of the scene blob failed (%d)" % rc)
    return buf, n


def broadcast_scene(scene, device, src=0, group=None, *, to_tensor=None, adopt=None, torch_device=None):
    """Broadcast a scene from rank ``src``; returns a scene valid on this rank.

    ``scene`` is the built ``horayzon_amd.Scene`` on ``src`` and ``None`` elsewhere.  The
    blob is received into a torch uint8 tensor that the adopted scene keeps alive
    (``hz_scene_adopt`` wraps it, no copy).  ``to_tensor(scene) -> (uint8 tensor, nbytes)``
    and ``adopt(tensor, nbytes) -> scene`` default to the HIP versions; the gloo test passes
    CPU stand-ins so that this very function runs without a GPU."""
    import torch.distributed as dist
    from . import _lib
    rank = dist.get_rank(group)
    if torch_device is None:
        torch_device = "cuda:%d" % device
    buf, n = (None, 0)
    if rank == src:
        buf, n = (to_tensor or (lambda s: _hip_blob_tensor(s, device)))(scene)
    buf = broadcast_blob(buf, n, torch_device, src=src, group=group)
    if rank == src:
        return scene
    if adopt is not None:
        return adopt(buf, int(buf.numel()))
    return _lib.Scene.adopt(buf.data_ptr(), int(buf.numel()), device, keepalive=buf)


def _host_staged(t, group):
    """gloo moves only CPU tensors for gather / all_gather: a CUDA tensor on a gloo group (several ranks sharing one
    GPU in a test) goes through host memory; on RCCL the tensor is used as it is."""
    import torch.distributed as dist
    return t.cpu() if (t.is_cuda and dist.get_backend(group) == "gloo") else t


def gather_rows(local, slabs, dst=0, group=None):
    """Gather per-rank row slabs (torch tensors, leading axis = rows of the slab; ``None`` for an
    empty slab) into the full array on ``dst``.  CUDA tensors (RCCL) and CPU tensors (gloo).
    Meant for small per-cell outputs (SVF): every rank sends max-slab-rows rows."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    max_rows = max(max(e - b for b, e in slabs), 1)
    if local is None:
        raise ValueError("gather_rows needs a (possibly 0-row) tensor on every rank")
    tail = tuple(local.shape[1:])
    dev = local.device
    local = _host_staged(local, group)
    padded = torch.zeros((max_rows,) + tail, dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    out = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    try:
        dist.gather(padded, out, dst=dst, group=group)
    except (RuntimeError, NotImplementedError):      # a backend without gather: all_gather, keep dst's copy
        allv = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(allv, padded, group=group)
        out = allv if rank == dst else None
    if rank != dst:
        return None
    return torch.cat([out[r][:slabs[r][1] - slabs[r][0]] for r in range(world)], dim=0).to(dev)


def sharded_rows(mask, compute, *, sync=None, dst=0, group=None, gather=True):
    """Per-rank body of a row-sharded job: split the inner-domain rows by ``row_slabs(mask,
    world)``, run ``compute(begin, end) -> tensor[end - begin, ...]`` on this rank's slab (no collective
    in there), then gather the per-rank results on ``dst``.

    ``sync()`` is called after ``compute`` before the clock stops (torch.cuda.synchronize on GPUs).
    Returns a dict: ``full`` (the gathered array on ``dst``, else None), ``slabs``, ``t_compute`` (this
    rank's seconds), ``t_ranks`` (all ranks' seconds), ``imbalance`` (slowest rank / mean rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    slabs = row_slabs(mask, world)
    b, e = slabs[rank]
    t0 = time.perf_counter()
    local = compute(b, e)
    if sync is not None:
        sync()
    t_compute = time.perf_counter() - t0
    t = _host_staged(torch.tensor([t_compute], dtype=torch.float64, device=local.device), group)
    ts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(ts, t, group=group)
    t_ranks = [float(x.item()) for x in ts]
    busy = [x for x, (sb, se) in zip(t_ranks, slabs) if se > sb]
    imbalance = max(busy) / (sum(busy) / len(busy)) if busy else 1.0
    full = gather_rows(local, slabs, dst=dst, group=group) if gather else None
    return dict(full=full, slabs=slabs, t_compute=t_compute, t_ranks=t_ranks, imbalance=imbalance)
