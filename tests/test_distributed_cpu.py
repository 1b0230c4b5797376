"""CPU, world_size 2, gloo: the partition + single all-gather that carries the N > 1 path (phoskintime_amd/distributed.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from phoskintime_amd.distributed import (shard_bounds, all_gather_replicas, sharded_map, shared_seed, all_gather_with_status, interleaved_rows, cost_order,
                                         all_gather_interleaved, all_gather_interleaved_with_status, sharded_map_rows)


def test_shard_bounds_cover_exactly_once():
    for total in (0, 1, 7, 8, 9, 65536, 25728):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            assert sum(b - a for a, b in spans) == total
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def fn(lo, hi):          # stand-in for the kernel: a per-replica result that encodes the global index
            idx = torch.arange(lo, hi, dtype=torch.float64)
            return torch.stack([idx, idx * idx + rank * 0.0], dim=1)
        full = sharded_map(fn, total)
        ok = full.shape == (total, 2) and torch.equal(full[:, 0], torch.arange(total, dtype=torch.float64)) \
            and torch.equal(full[:, 1], full[:, 0] ** 2)
        lo, hi = shard_bounds(total, rank, world)
        one = all_gather_replicas(torch.full((hi - lo,), float(rank), dtype=torch.float64), total)
        ok = ok and one.shape == (total,) and float(one.sum()) == float(sum(r * (shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0]) for r in range(world)))
        # values + status flags through one collective
        vals = torch.arange(lo, hi, dtype=torch.float64) * 0.5
        flags = torch.tensor([(i % 3) * 2 for i in range(lo, hi)], dtype=torch.int32)
        v, st = all_gather_with_status(vals, flags, total)
        ok = ok and torch.equal(v, torch.arange(total, dtype=torch.float64) * 0.5) and st.dtype == torch.int32 \
            and st.tolist() == [(i % 3) * 2 for i in range(total)]
        # seed=None must still give every rank the same Morris design (each rank used to draw its own)
        from phoskintime_amd.sensitivity import morris
        s = shared_seed(None)
        d = morris.draw(5, 3, 4, seed=s)
        sig = torch.tensor([float(s), float(d.base.sum()), float(d.sign.sum()), float((d.rank * [1, 2, 3, 4, 5]).sum())], dtype=torch.float64)
        both = [torch.empty_like(sig) for _ in range(world)]
        dist.all_gather(both, sig)
        ok = ok and all(torch.equal(b, both[0]) for b in both) and shared_seed(17) == 17
        # rows-sharded least-squares driver: the partition / packing / gather around the (here: faked) per-rank fit
        import numpy as np
        from phoskintime_amd.paramest import multistart as ms
        calls = []

        def fake_fit(model, n, t, P0, y0, target, sigma=None, lam=0.0, bounds=None, **kw):
            calls.append((P0.shape[0], np.asarray(bounds[0]).shape, np.asarray(lam).shape))
            R, P = P0.shape
            return ms.RowsFit(p=P0 * 2.0, cost=P0.sum(axis=1), r=np.tile(P0[:, :1], (1, 4)), JTJ=P0[:, :, None] * P0[:, None, :], n_iter=3, n_solves=7, n_launches=2)
        ms.fit_rows_batch = fake_fit
        R_ = max(total, 1)
        P0 = np.arange(R_ * 3, dtype=float).reshape(R_, 3)
        fit = ms.fit_rows_sharded("distmod", 1, [0.0, 1.0], P0, np.ones(3), np.zeros(4), lam=np.zeros(R_), bounds=(np.zeros((R_, 3)), np.ones(3)))
        ok = ok and np.array_equal(fit.p, P0 * 2.0) and np.array_equal(fit.cost, P0.sum(axis=1)) and fit.JTJ.shape == (R_, 3, 3) \
            and np.array_equal(fit.JTJ[-1], np.outer(P0[-1], P0[-1])) and fit.r.shape == (R_, 4)
        mine = interleaved_rows(R_, rank, world).numel()
        ok = ok and ((mine == 0 and not calls) or (calls and calls[0] == (mine, (mine, 3), (mine,))))      # per-row args sliced (interleaved rows), shared passed on
        # interleaved partition: the gathered result equals the single-process one BIT FOR BIT, with and without a cost
        # order; rows independent => fn(arange(total)) is the single-process answer
        g = torch.Generator().manual_seed(5)
        data = torch.rand((total, 3), dtype=torch.float64, generator=g)
        kernel = lambda rows: torch.stack([data[rows, 0] * data[rows, 1], torch.sin(data[rows, 2]) + rows.to(torch.float64)], dim=1)
        want = kernel(torch.arange(total))
        ok = ok and torch.equal(sharded_map_rows(kernel, total), want)
        cost = data[:, 2].clone()
        ok = ok and torch.equal(sharded_map_rows(kernel, total, cost=cost), want)
        rows_all = [interleaved_rows(total, r, world, cost_order(cost)) for r in range(world)]
        ok = ok and sorted(torch.cat(rows_all).tolist()) == list(range(total)) and abs(rows_all[0].numel() - rows_all[-1].numel()) <= 1
        if total >= 4:                # dealt in decreasing cost: the ranks' cost sums differ by less than the largest single cost
            sums = [float(cost[r_].sum()) for r_ in rows_all]
            ok = ok and abs(sums[0] - sums[1]) <= float(cost.max()) + 1e-12
        mine_rows = interleaved_rows(total, rank, world)
        v2, st2 = all_gather_interleaved_with_status(want[mine_rows], (mine_rows % 5).to(torch.int32), total)
        ok = ok and torch.equal(v2, want) and st2.tolist() == [i % 5 for i in range(total)]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 11, 1])
def test_sharded_map_gloo_world2(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_interleaved_partition_single_process():
    for total in (0, 1, 7, 8, 25728):
        for world in (1, 2, 3, 8):
            rows = [interleaved_rows(total, r, world) for r in range(world)]
            assert sorted(torch.cat(rows).tolist()) == list(range(total))
            assert max(x.numel() for x in rows) - min(x.numel() for x in rows) <= 1
    c = torch.tensor([0.5, 3.0, 3.0, 0.1, 9.0])
    assert cost_order(c).tolist() == [4, 1, 2, 0, 3]                      # decreasing, stable
    x = torch.arange(5.0, dtype=torch.float64)
    assert torch.equal(all_gather_interleaved(x, 5), x)
    o = cost_order(c)
    assert torch.equal(all_gather_interleaved(x[o], 5, o), x)             # single rank: local order = the cost order; back in global order
    assert torch.equal(sharded_map_rows(lambda r: r.to(torch.float64) * 2, 6, cost=torch.rand(6)), torch.arange(6, dtype=torch.float64) * 2)
    with pytest.raises(ValueError):
        interleaved_rows(10, 2, 2)


def test_single_rank_passthrough():
    assert shared_seed(None) is None and shared_seed(5) == 5
    v, st = all_gather_with_status(torch.arange(3.0, dtype=torch.float64), torch.tensor([0, 2, 4], dtype=torch.int32), 3)
    assert v.tolist() == [0.0, 1.0, 2.0] and st.tolist() == [0, 2, 4]
    x = torch.arange(5.0)
    assert all_gather_replicas(x, 5) is x
    assert torch.equal(sharded_map(lambda lo, hi: torch.arange(lo, hi, dtype=torch.float64), 4), torch.arange(4, dtype=torch.float64))
