"""world_size-2 gloo test of the multi-GPU sharding plumbing (row slabs, blob broadcast
as bytes, final gather).  On CPU there is no HIP compute, so the per-slab computation is
done by the oracle -- here only as the stand-in that makes the gathered result checkable
against the unsharded one."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from horayzon_amd.dist import gather_rows, row_slabs
    from oracle import oracle as orc
    from tests import cases
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        g = cases.rough_terrain(40, 36, seed=4, offset=3)
        kw = cases.grid_kwargs(g)
        rng = np.random.default_rng(0)
        mask = (rng.random(kw["vec_norm"].shape[:2]) > 0.3).astype(np.uint8)
        mask[:10] = 0                                   # unbalanced: slabs follow the mask
        # "scene" broadcast: rank 0 owns the vertex bytes, the others receive them
        blob = torch.from_numpy(kw["vert_grid"].copy()) if rank == 0 else torch.empty(kw["vert_grid"].size)
        dist.broadcast(blob, src=0)
        kw["vert_grid"] = blob.numpy()
        slabs = row_slabs(mask, world)
        b, e = slabs[rank]
        local, _ = orc.horizon_gridded(**kw, dist_search=1.0, azim_num=12, elev_ang_low_lim=-60.0,
                                       mask=mask, hori_fill=-2.0, rows=(b, e), slab_only=True)
        full = gather_rows(torch.from_numpy(local), slabs, dst=0)
        if rank == 0:
            ref, _ = orc.horizon_gridded(**kw, dist_search=1.0, azim_num=12, elev_ang_low_lim=-60.0,
                                         mask=mask, hori_fill=-2.0)
            q.put(("ok", bool(np.array_equal(full.numpy(), ref)), slabs))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_equals_unsharded_gloo():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    tag, equal, slabs = q.get(timeout=10)
    assert tag == "ok" and equal
    assert slabs[0][1] > 17                             # mask-balanced, not an even row split
