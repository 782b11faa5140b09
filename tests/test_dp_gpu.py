"""Multi-rank data-parallel path rehearsed on ONE GPU: two processes share the
device, gradients travel over gloo (host-staged), everything else -- sharding,
buckets, bucket events, the mean, the optimiser step -- is the code the RCCL
ranks run.  Checked against one process that runs both shards itself."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu

BATCH, H = 4, 32


def _graph():
    import torch_ref
    return torch_ref.unet_graph(H, H, 3, 32, 2, True)


def _data():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((BATCH, 3, H, H)).astype(np.float32)
    m = (rng.random((BATCH, H, H)) < 0.3).astype(np.float32)
    return x, m


def _make_net(bucket_bytes):
    import neural_net as nn
    import torch_ref
    net = nn.NeuralNet()
    net.set_seed(3)
    net.create(torch_ref.to_layer_descs(nn, _graph()), nn.LossFunction.DICE_LOSS)
    net.reserve(BATCH)
    net.set_bucket_bytes(bucket_bytes)
    return net


def _rank_main(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        from unet_amd import capi, dp
        torch.cuda.set_device(0)
        capi.unet_set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)
        net = _make_net(64 << 10)          # small buckets: several per step
        trainer = dp.DataParallel(net, dev, bucket_bytes=64 << 10)
        trainer.broadcast_parameters()
        x, m = _data()
        lo, hi = dp.shard_range(BATCH, rank, world)
        xs = torch.from_numpy(x[lo:hi]).to(dev)
        ms = torch.from_numpy(m[lo:hi]).to(dev)
        p0 = net.get_parameters()
        net.forward_backward_device(xs.data_ptr(), ms.data_ptr(), hi - lo)
        trainer.reduce_gradients()
        net.synchronize()
        g = net.get_gradients()
        net.step(1e-3)
        net.synchronize()
        p1 = net.get_parameters()
        nb = len(trainer.buckets)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, True, (np.asarray(p0), np.asarray(g), np.asarray(p1), nb)))
    except Exception as e:  # report instead of hanging the parent
        import traceback
        q.put((rank, False, traceback.format_exc() + str(e)))


def test_two_ranks_on_one_gpu_match_single_process():
    import torch
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, ok, payload = q.get(timeout=240)
        assert ok, f"rank {r}: {payload}"
        res[r] = payload
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    p0_a, g_a, p1_a, nb = res[0]
    p0_b, g_b, p1_b, _ = res[1]
    assert nb >= 3, "the test wants several gradient buckets"
    assert np.array_equal(p0_a, p0_b), "parameters were not broadcast"
    assert np.array_equal(g_a, g_b), "ranks hold different reduced gradients"
    assert np.array_equal(p1_a, p1_b), "replicas diverged after the optimiser step"

    # one process, both shards in turn, gradients averaged on the host
    from unet_amd import dp
    dev = torch.device("cuda", 0)
    net = _make_net(64 << 10)
    net.set_parameters(p0_a)
    x, m = _data()
    grads = []
    for r in range(2):
        lo, hi = dp.shard_range(BATCH, r, 2)
        xs = torch.from_numpy(x[lo:hi]).to(dev)
        ms = torch.from_numpy(m[lo:hi]).to(dev)
        net.forward_backward_device(xs.data_ptr(), ms.data_ptr(), hi - lo)
        net.synchronize()
        grads.append(np.asarray(net.get_gradients(), dtype=np.float64))
    want = (grads[0] + grads[1]) / 2
    scale = np.abs(want).max()
    assert np.abs(g_a - want).max() <= 1e-6 * scale + 1e-12, np.abs(g_a - want).max() / scale
    assert np.isfinite(p1_a).all() and not np.array_equal(p1_a, p0_a)


def _rccl_rank(port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        from unet_amd import capi, dp
        torch.cuda.set_device(0)
        capi.unet_set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        x, m = _data()
        xs, ms = torch.from_numpy(x).to(dev), torch.from_numpy(m).to(dev)
        plain = _make_net(64 << 10)
        p0 = plain.get_parameters()
        plain.forward_backward_device(xs.data_ptr(), ms.data_ptr(), BATCH)
        plain.step(1e-3)
        plain.synchronize()
        want_g, want_p = np.asarray(plain.get_gradients()), np.asarray(plain.get_parameters())
        net = _make_net(64 << 10)
        net.set_parameters(p0)
        trainer = dp.DataParallel(net, dev, bucket_bytes=64 << 10, force_collectives=True)
        trainer.broadcast_parameters()
        for _ in range(2):   # twice: the second pass reuses the bucket events and the communication stream
            net.set_parameters(p0)
            trainer.train_step(xs.data_ptr(), ms.data_ptr(), BATCH, 1e-3)
            net.synchronize()
            torch.cuda.synchronize()
        got_g, got_p = np.asarray(net.get_gradients()), np.asarray(net.get_parameters())
        dist.destroy_process_group()
        q.put((True, (want_g, got_g, want_p, got_p, len(trainer.buckets))))
    except Exception as e:
        import traceback
        q.put((False, traceback.format_exc() + str(e)))


def test_rccl_transport_one_rank():
    """The RCCL code path itself (communication stream, bucket events, asynchronous all-reduce with
    AVG, the wait before the optimiser step) on a one-rank communicator: the mean over one rank must
    leave the gradients and the update unchanged."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_rank, args=(port, q))
    p.start()
    ok, payload = q.get(timeout=240)
    p.join(timeout=60)
    assert ok, payload
    want_g, got_g, want_p, got_p, nb = payload
    assert nb >= 3
    # the second step starts from the first step's Adam state, so compare gradients (state-free)
    assert np.array_equal(want_g, got_g)
    assert np.isfinite(got_p).all()
