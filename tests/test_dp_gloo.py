"""Data-parallel glue on CPU: world_size 2 over gloo (the same code path the
GPU ranks take over RCCL, minus the streams)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from unet_amd import dp  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, fn, *args):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(r, world, port, fn, q, args)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        r, ok, payload = q.get(timeout=120)
        assert ok, f"rank {r}: {payload}"
        results[r] = payload
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def _entry(rank, world, port, fn, q, args):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        out = fn(rank, world, *args)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, True, out))
    except Exception as e:  # report instead of hanging the parent
        import traceback
        q.put((rank, False, traceback.format_exc() + str(e)))


def test_shard_range_partitions_the_batch():
    for total in (0, 1, 7, 16, 64, 129):
        for world in (1, 2, 3, 8):
            seen = []
            sizes = []
            for r in range(world):
                lo, hi = dp.shard_range(total, r, world)
                assert 0 <= lo <= hi <= total
                seen += list(range(lo, hi))
                sizes.append(hi - lo)
            assert seen == list(range(total))
            assert max(sizes) - min(sizes) <= 1


def _allreduce_case(rank, world):
    n = 1000
    rng = np.random.default_rng(100 + rank)
    flat = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
    mine = flat.clone()
    # buckets in "backward order": the tail of the vector first, ragged sizes, one empty
    buckets = [(700, 300, None), (700, 0, None), (123, 577, None), (0, 123, None)]
    works = dp.allreduce_mean_(flat, buckets)
    assert works == []
    return mine.numpy(), flat.numpy()


def test_allreduce_mean_world2():
    res = _run(2, _allreduce_case)
    want = (res[0][0].astype(np.float64) + res[1][0].astype(np.float64)) / 2
    for r in (0, 1):
        np.testing.assert_allclose(res[r][1], want, rtol=1e-6, atol=1e-7)
    # both ranks hold bit-identical reduced gradients (replicas stay in lockstep)
    assert np.array_equal(res[0][1], res[1][1])


def _conv_grad_case(rank, world, batch):
    """Each rank: weight gradient of 0.5*mean((conv(x)-t)^2) over ITS shard with
    the CPU oracle, then the bucketed mean over ranks."""
    import oracle
    rng = np.random.default_rng(7)
    ci, co, h = 3, 4, 8
    x = rng.standard_normal((batch, ci, h, h)).astype(np.float32)
    t = rng.standard_normal((batch, co, h, h)).astype(np.float32)
    w = (rng.standard_normal(ci * co * 9 + co) * 0.2).astype(np.float32)
    lo, hi = dp.shard_range(batch, rank, world)
    xs, ts = x[lo:hi], t[lo:hi]
    y = oracle.convolve(xs, w, co, 3, 1, 1)
    dy = ((y - ts) / y.size).astype(np.float32)
    _, dw = oracle.convolve_backward(xs, dy, w, 3, 1, 1)
    flat = torch.from_numpy(np.ascontiguousarray(dw, dtype=np.float32).copy())
    nw = ci * co * 9
    dp.allreduce_mean_(flat, [(nw, co, None), (0, nw, None)])
    # full-batch gradient for comparison
    yf = oracle.convolve(x, w, co, 3, 1, 1)
    _, dwf = oracle.convolve_backward(x, ((yf - t) / yf.size).astype(np.float32), w, 3, 1, 1)
    return flat.numpy(), np.asarray(dwf)


def test_sharded_gradient_mean_equals_full_batch_gradient():
    res = _run(2, _conv_grad_case, 6)
    for r in (0, 1):
        got, want = res[r]
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-7)


def _broadcast_case(rank, world):
    flat = torch.full((17,), float(rank + 1))
    dist.broadcast(flat, 0)
    return flat.numpy()


def test_parameter_broadcast_world2():
    res = _run(2, _broadcast_case)
    assert np.array_equal(res[0], res[1]) and res[1][0] == 1.0


def test_world1_is_a_no_op():
    res = _run(1, _allreduce_case)
    assert np.array_equal(res[0][0], res[0][1])
