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


def row_slabs(mask_or_rows, world_size, cost=None):
    """Split rows into ``world_size`` contiguous slabs balanced by work.  The work of a row is the number of
    cells to compute (``mask == 1``; accepts a 2-D mask or a row count) or, when ``cost`` (one non-negative
    weight per row, e.g. from ``estimate_row_cost``) is given, that weight: rows over rough terrain cost more rays
    and node visits than rim rows whose rays leave the DEM early.  Returns [(begin, end)] * world_size; slabs may
    be empty when rows < world_size."""
    if cost is not None:
        w = np.asarray(cost, np.float64)
        if w.ndim != 1 or (w < 0).any() or not np.isfinite(w).all():
            raise ValueError("'cost' must be one finite non-negative weight per row")
        if np.ndim(mask_or_rows) != 0 and np.asarray(mask_or_rows).shape[0] != w.shape[0]:
            raise ValueError("'cost' and the mask disagree on the number of rows")
    elif np.ndim(mask_or_rows) == 0:
        w = np.ones(int(mask_or_rows), np.float64)
    else:
        w = (np.asarray(mask_or_rows) == 1).sum(axis=1).astype(np.float64)
    n = w.shape[0]
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        b = int(np.searchsorted(cum, target, side="left"))
        # the boundary that leaves the smaller error (searchsorted alone always rounds up)
        if b > 0 and b <= n and abs(cum[b - 1] - target) <= abs(cum[min(b, n)] - target):
            b -= 1
        bounds.append(min(max(b, bounds[-1]), n))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def sample_rows(n_rows, n_samples):
    """``n_samples`` row indices spread evenly over [0, n_rows) (centres of equal strata), without duplicates."""
    n_samples = max(1, min(int(n_samples), int(n_rows)))
    idx = ((np.arange(n_samples) + 0.5) * n_rows / n_samples).astype(np.int64)
    return np.unique(np.clip(idx, 0, n_rows - 1))


def estimate_row_cost(mask_or_rows, probe, *, samples=32, group=None, refine=3, refine_ratio=1.5, min_gap=16):
    """Per-row cost weights for ``row_slabs(..., cost=...)`` from a cheap sampled pre-pass.

    ``probe(row) -> float`` measures the cost of ONE inner-domain row (e.g. weighted node visits / triangle tests /
    rays of a ``count_work`` call on a sample of the row's tiles; any unit).  The ``samples`` probe rows are
    split over the ranks of ``group`` (rank r takes samples r, r + world, ...; one small all_reduce joins them; no
    process group: this process probes all of them), converted to a cost per cell, interpolated linearly between the
    sampled rows and multiplied by every row's cell count.  ``refine`` rounds of bisection follow: wherever two
    neighbouring samples differ by more than ``refine_ratio`` in cost per cell (and are more than ``min_gap`` rows apart)
    the row half way between them is probed as well -- a cliff between a plain and high relief is a band of rows that
    costs ten times its surroundings, and evenly spaced samples step over it (bench.py --plain-fraction, round 4).
    Deterministic on every rank: all ranks derive the same probe rows and the same slabs without exchanging them."""
    if np.ndim(mask_or_rows) == 0:
        cells = np.ones(int(mask_or_rows), np.float64)
    else:
        cells = (np.asarray(mask_or_rows) == 1).sum(axis=1).astype(np.float64)
    n = cells.shape[0]
    rank, world, dist = 0, 1, None
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(group), dist.get_world_size(group)
    except ImportError:
        dist = None

    def measure(rows):
        """cost of the given rows, the probes split over the ranks"""
        local = np.zeros(rows.shape[0], np.float64)
        for k in range(rank, rows.shape[0], world):
            local[k] = float(probe(int(rows[k])))
        if world > 1:
            import torch
            t = torch.from_numpy(local)
            if dist.get_backend(group) != "gloo":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)     # disjoint supports: the sum joins the shares
            local = t.cpu().numpy()
        return local

    rows = sample_rows(n, samples)
    cost = measure(rows)
    for _ in range(max(int(refine), 0)):
        per_cell = cost / np.maximum(cells[rows], 1.0)
        new = []
        for a in range(rows.shape[0] - 1):
            lo, hi = sorted((per_cell[a], per_cell[a + 1]))
            if rows[a + 1] - rows[a] > min_gap and hi > 0 and (lo <= 0 or hi / lo > refine_ratio):
                new.append((int(rows[a]) + int(rows[a + 1])) // 2)
        new = np.array(sorted(set(new) - set(rows.tolist())), np.int64)
        if new.size == 0:
            break
        c_new = measure(new)
        order = np.argsort(np.concatenate([rows, new]), kind="stable")
        rows = np.concatenate([rows, new])[order]
        cost = np.concatenate([cost, c_new])[order]
    per_cell = cost / np.maximum(cells[rows], 1.0)
    ok = cells[rows] > 0
    if not ok.any() or not (per_cell[ok] > 0).any():
        return cells                                              # nothing measured: fall back to the cell count
    dense = np.interp(np.arange(n), rows[ok], per_cell[ok])
    return dense * cells


def predicted_imbalance(slabs, cost):
    """max / mean of the slabs' summed cost (over the non-empty slabs), the figure ``sharded_rows`` measures as
    slowest rank / mean rank."""
    c = np.asarray(cost, np.float64)
    w = [float(c[b:e].sum()) for b, e in slabs if e > b]
    return max(w) / (sum(w) / len(w)) if w and sum(w) > 0 else 1.0


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


class _DeviceBytes:
    """A raw HBM range as an object torch can wrap without copying (__cuda_array_interface__, version 2)."""

    def __init__(self, ptr, nbytes, owner=None):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner


def device_bytes_tensor(ptr, nbytes, device, owner=None):
    """uint8 torch tensor over [ptr, ptr + nbytes) of GPU ``device`` -- no copy; ``owner`` is kept alive with it."""
    import torch
    t = torch.as_tensor(_DeviceBytes(ptr, nbytes, owner), device="cuda:%d" % device)
    if t.data_ptr() != int(ptr) or t.numel() != int(nbytes):
        raise RuntimeError("torch copied the device range instead of wrapping it")
    t._hz_owner = owner
    return t


def device_bytes_or_copy(ptr, nbytes, device, owner=None):
    """(uint8 tensor over the range, True) -- or, if torch cannot wrap it, (a torch-owned copy of the range, False)."""
    try:
        return device_bytes_tensor(ptr, nbytes, device, owner=owner), True
    except Exception:
        import ctypes as C
        import torch
        from . import _lib
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda:%d" % device)
        path = next((l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l), "libamdhip64.so")
        rc = C.CDLL(path).hipMemcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(int(ptr)), C.c_size_t(int(nbytes)), 3)
        if rc != 0:
            raise _lib.HorayzonHipError("hipMemcpy of a device range failed (%d)" % rc)
        return buf, False


last_broadcast_zero_copy = None      # True / False after a broadcast on the source rank (bench.py reports it)


def _hip_blob_tensor(scene, device):
    """The scene's blob AS a torch uint8 CUDA tensor: the send buffer of the broadcast is the blob allocation itself
    (no second copy of a 17.7 GB blob on the source rank).  If torch cannot wrap the range (an unexpected build of
    torch), the blob is copied into a torch tensor as in round 2 -- slower, same result."""
    global last_broadcast_zero_copy
    p, n = scene.blob()
    t, last_broadcast_zero_copy = device_bytes_or_copy(p, n, device, owner=scene)
    return t, n


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
    """Gather per-rank row slabs (torch tensors, leading axis = rows of the slab; a 0-row tensor for an
    empty slab) into the full array on ``dst``.  CUDA tensors (RCCL) and CPU tensors (gloo).
    Every rank sends exactly its own rows to ``dst`` (point-to-point; rounds 2-3 padded every slab to the
    largest one for a `gather` collective -- with cost-balanced slabs the sizes differ by design)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if local is None:
        raise ValueError("gather_rows needs a (possibly 0-row) tensor on every rank")
    tail = tuple(local.shape[1:])
    dev = local.device
    local = _host_staged(local, group).contiguous()
    if local.shape[0] != slabs[rank][1] - slabs[rank][0]:
        raise ValueError("rank %d holds %d rows, its slab has %d" % (rank, local.shape[0], slabs[rank][1] - slabs[rank][0]))
    # `dst` and the slab index r are ranks OF THE GROUP; torch's point-to-point calls take GLOBAL ranks
    def glob(r):
        return r if group is None else dist.get_global_rank(group, r)
    if rank != dst:
        if local.shape[0] > 0:
            dist.send(local, dst=glob(dst), group=group)
        return None
    parts = []
    for r in range(world):
        rows = slabs[r][1] - slabs[r][0]
        if r == dst:
            parts.append(local)
        elif rows > 0:
            buf = torch.empty((rows,) + tail, dtype=local.dtype, device=local.device)
            dist.recv(buf, src=glob(r), group=group)
            parts.append(buf)
    return torch.cat(parts, dim=0).to(dev)


def sharded_rows(mask, compute, *, sync=None, dst=0, group=None, gather=True, cost=None):
    """Per-rank body of a row-sharded job: split the inner-domain rows by ``row_slabs(mask,
    world, cost)``, run ``compute(begin, end) -> tensor[end - begin, ...]`` on this rank's slab (no collective
    in there), then gather the per-rank results on ``dst``.  ``mask`` may be a 2-D mask or a row count.

    ``sync()`` is called after ``compute`` before the clock stops (torch.cuda.synchronize on GPUs).
    Returns a dict: ``full`` (the gathered array on ``dst``, else None), ``slabs``, ``t_compute`` (this
    rank's seconds), ``t_ranks`` (all ranks' seconds), ``imbalance`` (slowest rank / mean rank, measured) and
    ``imbalance_predicted`` (the same ratio of the slabs' cost weights; None without ``cost``)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    slabs = row_slabs(mask, world, cost)
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
    return dict(full=full, slabs=slabs, t_compute=t_compute, t_ranks=t_ranks, imbalance=imbalance,
                imbalance_predicted=predicted_imbalance(slabs, cost) if cost is not None else None)
