"""world_size-2 gloo test of the multi-GPU code path: the SAME functions bench.py runs on RCCL --
``dist.broadcast_scene`` (blob bytes from the building rank, adopted by the receivers) and
``dist.sharded_rows`` (mask-balanced row slabs, per-rank compute with no collective, timing exchange,
final gather) -- with CPU tensors.  There is no HIP compute on CPU, so the per-slab computation is done
by the oracle: only as the stand-in that makes the gathered result checkable against the unsharded one."""
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
    from horayzon_amd.dist import broadcast_scene, sharded_rows
    from oracle import oracle as orc
    from tests import cases
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        g = cases.rough_terrain(40, 36, seed=4, offset=3)
        kw = cases.grid_kwargs(g)
        rng = np.random.default_rng(0)
        mask = (rng.random(kw["vec_norm"].shape[:2]) > 0.3).astype(np.uint8)
        mask[:10] = 0                                   # unbalanced: slabs follow the mask
        vec_tilt, *_ = cases.terrain_inputs(g)
        # the "scene": rank 0 owns the vertex bytes (stand-in for the LBVH blob), the others adopt what arrives
        verts0 = kw["vert_grid"].copy()
        scene = verts0 if rank == 0 else None
        got = broadcast_scene(
            scene, 0, src=0, torch_device="cpu",
            to_tensor=lambda s: (torch.from_numpy(s.view(np.uint8)), s.nbytes),
            adopt=lambda buf, n: buf.numpy()[:n].view(np.float32))
        kw["vert_grid"] = np.ascontiguousarray(got)
        assert np.array_equal(kw["vert_grid"], verts0)

        def compute(b, e):          # SVF of the slab, horizon never leaves the rank
            if e <= b:
                return torch.empty((0, mask.shape[1]), dtype=torch.float32)
            h, azim = orc.horizon_gridded(**kw, dist_search=1.0, azim_num=12, elev_ang_low_lim=-60.0,
                                          mask=mask, hori_fill=-2.0, rows=(b, e), slab_only=True)
            return torch.from_numpy(orc.sky_view_factor(azim, h, np.ascontiguousarray(vec_tilt[b:e])))

        res = sharded_rows(mask, compute, dst=0)
        # cost-balanced variant: a sampled pre-pass whose probe rows are split over the two ranks (one all_reduce
        # joins the shares); the probe here is a stand-in (row index as cost per cell), the machinery is the product's
        from horayzon_amd.dist import estimate_row_cost, row_slabs, predicted_imbalance
        probed = []
        cost = estimate_row_cost(mask, lambda r: (probed.append(r), (1.0 + r) * float(mask[r].sum()))[1], samples=9)
        assert 4 <= len(probed) <= 5 and cost.shape == (mask.shape[0],) and np.all(cost[:10] == 0.0)
        res_c = sharded_rows(mask, compute, dst=0, cost=cost)
        assert res_c["slabs"] == row_slabs(mask, world, cost) and res_c["slabs"] != res["slabs"]
        assert res_c["imbalance_predicted"] == predicted_imbalance(res_c["slabs"], cost) < predicted_imbalance(res["slabs"], cost)
        if rank == 0:
            assert np.array_equal(res_c["full"].numpy(), res["full"].numpy())     # any partition, same result
        if rank == 0:
            h, azim = orc.horizon_gridded(**kw, dist_search=1.0, azim_num=12, elev_ang_low_lim=-60.0,
                                          mask=mask, hori_fill=-2.0)
            ref = orc.sky_view_factor(azim, h, vec_tilt)
            q.put(("ok", bool(np.array_equal(res["full"].numpy(), ref)), res["slabs"], res["t_ranks"],
                   res["imbalance"]))
        else:
            assert res["full"] is None
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
    tag, equal, slabs, t_ranks, imbalance = q.get(timeout=10)
    assert tag == "ok" and equal
    assert slabs[0][1] > 17                             # mask-balanced, not an even row split
    assert len(t_ranks) == 2 and all(t > 0 for t in t_ranks) and 1.0 <= imbalance <= 2.0


def _subgroup_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from horayzon_amd.dist import gather_rows
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        grp = dist.new_group([1, 2])                    # every rank calls new_group; ranks 1 and 2 are its members
        if rank in (1, 2):
            gr = dist.get_rank(grp)                     # 0 / 1 inside the group = global 1 / 2
            slabs = [(0, 3), (3, 5)]
            local = torch.full((slabs[gr][1] - slabs[gr][0], 4), float(10 + gr))
            full = gather_rows(local, slabs, dst=0, group=grp)
            if gr == 0:
                q.put(("ok", full.tolist()))
            else:
                assert full is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_rows_on_a_sub_group():
    """gather_rows addresses its peers by rank OF THE GROUP; torch's send / recv take global ranks: on a group
    that is not the world -- ranks 1 and 2 of three -- the rows must still arrive at the group's rank 0."""
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    tag, full = q.get(timeout=10)
    assert tag == "ok" and full == [[10.0] * 4] * 3 + [[11.0] * 4] * 2


def test_cost_balanced_slabs():
    from horayzon_amd.dist import row_slabs, estimate_row_cost, predicted_imbalance, sample_rows
    cost = np.concatenate([np.ones(60), 4.0 * np.ones(40)])         # the last 40 rows cost four times as much
    by_cost, by_cells = row_slabs(100, 4, cost), row_slabs(100, 4)
    assert by_cells == [(0, 25), (25, 50), (50, 75), (75, 100)]
    assert by_cost[0][0] == 0 and by_cost[-1][1] == 100 and all(by_cost[i][1] == by_cost[i + 1][0] for i in range(3))
    assert predicted_imbalance(by_cost, cost) < 1.05 < 1.5 < predicted_imbalance(by_cells, cost)
    assert list(sample_rows(10, 4)) == [1, 3, 6, 8] and list(sample_rows(3, 8)) == [0, 1, 2]
    m = np.ones((100, 7), np.uint8); m[:20] = 0                      # masked rows cost nothing whatever the probe says
    c = estimate_row_cost(m, lambda r: 7.0 * (1.0 + (r >= 60) * 3.0), samples=10)     # no process group: all probes here
    assert np.all(c[:20] == 0) and np.allclose(c[20:50], 7.0) and np.allclose(c[70:], 28.0)
    assert np.array_equal(estimate_row_cost(5, lambda r: 0.0), np.ones(5))           # nothing measured: cell count
    with pytest.raises(ValueError):
        row_slabs(10, 2, cost=np.ones(9) * -1)


def test_row_slabs_edge_cases():
    from horayzon_amd.dist import row_slabs
    assert row_slabs(10, 1) == [(0, 10)]
    assert row_slabs(3, 8)[-1][1] == 3 and sum(e - b for b, e in row_slabs(3, 8)) == 3
    m = np.zeros((12, 5), np.uint8); m[8:] = 1           # all the work in the last rows
    sl = row_slabs(m, 4)
    assert sl[0][0] == 0 and sl[-1][1] == 12 and all(sl[i][1] == sl[i + 1][0] for i in range(3))
    work = [int(m[b:e].sum()) for b, e in sl]
    assert max(work) - min(work) <= 5                    # balanced to within one row of cells


def test_cost_estimate_refines_across_a_narrow_expensive_band():
    """estimate_row_cost bisects where neighbouring samples differ strongly: a band of rows that costs 20 times its
    surroundings (a cliff between a plain and high relief) is resolved with a few more probes instead of being stepped over."""
    import numpy as np
    from horayzon_amd.dist import estimate_row_cost, row_slabs, predicted_imbalance
    n = 2000
    true = np.ones(n)
    true[700:790] = 20.0
    true[790:] = 2.0
    calls = []

    def probe(row):
        calls.append(row)
        return float(true[row])

    coarse = estimate_row_cost(n, probe, samples=32, refine=0)
    n_coarse = len(calls)
    fine = estimate_row_cost(n, probe, samples=32, refine=3)
    n_fine = len(calls) - n_coarse
    assert n_coarse == 32 and 32 < n_fine <= 32 + 24
    imb_coarse = predicted_imbalance(row_slabs(n, 4, coarse), true)
    imb_fine = predicted_imbalance(row_slabs(n, 4, fine), true)
    imb_cells = predicted_imbalance(row_slabs(n, 4), true)
    assert imb_fine < 1.12 and imb_fine < imb_coarse - 0.03 and imb_fine < imb_cells - 0.2, (imb_cells, imb_coarse, imb_fine)
