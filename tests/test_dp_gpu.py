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
        x, m = _data()
        if rank == 1:
            # this rank has already run a forward on parameters of its own: its packed bf16 weights and cached
            # inference coefficients are stale once the broadcast overwrites the parameter buffer
            net.set_parameters(np.asarray(net.get_parameters()) * 1.5 + 0.01)
            net.forward(x[:1], True)
        trainer.broadcast_parameters()
        y_after_bcast = np.asarray(net.forward(x[:1], True))
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
        q.put((rank, True, (np.asarray(p0), np.asarray(g), np.asarray(p1), nb, y_after_bcast)))
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

    p0_a, g_a, p1_a, nb, y_a = res[0]
    p0_b, g_b, p1_b, _, y_b = res[1]
    assert nb >= 3, "the test wants several gradient buckets"
    assert np.array_equal(p0_a, p0_b), "parameters were not broadcast"
    assert np.array_equal(y_a, y_b), "a rank kept packed weights / inference coefficients from before the broadcast"
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


def _rccl_rank(port, q, sync_bn, break_comm=False, bf16_gradients=False):
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
        if break_comm == "preflight":   # some rank cannot load RCCL: found out before the collective set-up
            dp.library_transport_ready = lambda *args, **kwargs: (False, "simulated: cannot open librccl.so.1")
        elif break_comm:   # the library's communicator cannot be created: torch.distributed takes over
            def no_comm(*args, **kwargs):
                raise RuntimeError("simulated: ncclCommInitRank failed")
            dp.init_comm = no_comm
        trainer = dp.DataParallel(net, dev, bucket_bytes=64 << 10, force_collectives=True, sync_bn=sync_bn,
                                  bf16_gradients=bf16_gradients)
        if break_comm:
            assert not trainer.native and trainer.comm is None and trainer.torch_device_reduce
        else:
            assert trainer.native and trainer.comm is not None
        trainer.broadcast_parameters()
        for _ in range(2):   # twice: the second pass reuses the bucket events and the communication stream
            net.set_parameters(p0)
            trainer.train_step(xs.data_ptr(), ms.data_ptr(), BATCH, 1e-3)
            net.synchronize()
            torch.cuda.synchronize()
        got_g, got_p = np.asarray(net.get_gradients()), np.asarray(net.get_parameters())
        nb = len(trainer.buckets)
        trainer.close()
        dist.destroy_process_group()
        q.put((True, (want_g, got_g, want_p, got_p, nb)))
    except Exception as e:
        import traceback
        q.put((False, traceback.format_exc() + str(e)))


def _run_one_rank(sync_bn, break_comm=False, bf16_gradients=False):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_rank, args=(port, q, sync_bn, break_comm, bf16_gradients))
    p.start()
    ok, payload = q.get(timeout=240)
    p.join(timeout=60)
    assert ok, payload
    return payload


def test_rccl_transport_one_rank():
    """The RCCL path of the C ABI itself (unet_comm_init_rank, the communication stream, bucket events,
    unet_allreduce_begin with the mean, unet_allreduce_wait before the optimiser step, unet_broadcast_f32) on
    a one-rank communicator: the mean over one rank must leave the gradients and the update unchanged."""
    want_g, got_g, want_p, got_p, nb = _run_one_rank(False)
    assert nb >= 3
    # the second step starts from the first step's Adam state, so compare gradients (state-free)
    assert np.array_equal(want_g, got_g)
    assert np.isfinite(got_p).all()


@pytest.mark.parametrize("how", ["preflight", "init"])
def test_torch_distributed_takes_over_when_the_library_communicator_fails(how):
    """RCCL not loadable on some rank, or unet_comm_init_rank failing (both simulated), must not end a
    data-parallel run: DataParallel then reduces the gradient buckets with torch.distributed on the device
    buffers; on one rank the step equals the plain one."""
    want_g, got_g, want_p, got_p, nb = _run_one_rank(False, break_comm=how)
    assert np.array_equal(want_g, got_g)
    assert np.isfinite(got_p).all()


def test_bf16_gradient_transport_one_rank():
    """unet_comm_set_bf16_gradients: every bucket is rounded to bf16, averaged by RCCL in bf16 and widened back.
    On one rank the mean is the identity, so the gradients must come out as the plain ones rounded to bf16
    (round to nearest even), bit for bit."""
    import torch
    want_g, got_g, want_p, got_p, nb = _run_one_rank(False, bf16_gradients=True)
    rounded = torch.from_numpy(want_g).bfloat16().float().numpy()
    assert np.array_equal(rounded, got_g)
    assert not np.array_equal(want_g, got_g)
    assert np.isfinite(got_p).all()


def test_sync_bn_one_rank():
    """Batch-norm statistics routed through the cross-rank sum (local sums in double, all-reduce, finalize on
    one row of global sums) on a one-rank communicator: same statistics, so the same gradients up to the
    rounding of a different summation order."""
    want_g, got_g, want_p, got_p, nb = _run_one_rank(True)
    scale = np.abs(want_g).max()
    assert np.abs(got_g.astype(np.float64) - want_g).max() <= 1e-5 * scale, np.abs(got_g - want_g).max() / scale
    assert np.isfinite(got_p).all()


def _two_gpu_rank(rank, world, port, q, sync_bn):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        from unet_amd import capi, dp
        torch.cuda.set_device(rank)
        capi.unet_set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        net = _make_net(64 << 10)
        trainer = dp.DataParallel(net, dev, bucket_bytes=64 << 10, sync_bn=sync_bn)
        trainer.broadcast_parameters()
        x, m = _data()
        lo, hi = dp.shard_range(BATCH, rank, world)
        xs, ms = torch.from_numpy(x[lo:hi]).to(dev), torch.from_numpy(m[lo:hi]).to(dev)
        p0 = np.asarray(net.get_parameters())
        trainer.train_step(xs.data_ptr(), ms.data_ptr(), hi - lo, 1e-3)
        net.synchronize()
        g, p1 = np.asarray(net.get_gradients()), np.asarray(net.get_parameters())
        trainer.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, True, (p0, g, p1)))
    except Exception as e:
        import traceback
        q.put((rank, False, traceback.format_exc() + str(e)))


@pytest.mark.parametrize("sync_bn", [False, True])
def test_two_ranks_over_rccl(sync_bn):
    """World size 2 over RCCL, one GPU per rank (skipped on a one-GPU box).  Without synchronised batch-norm
    statistics the reduced gradient is the mean of the two shard gradients; with them the two-rank step
    equals the single-process step on the whole batch."""
    import ctypes
    import torch
    import torch.multiprocessing as mp
    from unet_amd import capi, dp
    count = ctypes.c_int()
    capi.unet_device_count(ctypes.byref(count))
    if count.value < 2:
        pytest.skip("needs two GPUs")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_gpu_rank, args=(r, 2, port, q, sync_bn)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, ok, payload = q.get(timeout=240)
        assert ok, f"rank {r}: {payload}"
        res[r] = payload
    for p in procs:
        p.join(timeout=60)
    (p0_a, g_a, p1_a), (p0_b, g_b, p1_b) = res[0], res[1]
    assert np.array_equal(p0_a, p0_b) and np.array_equal(g_a, g_b) and np.array_equal(p1_a, p1_b)
    dev = torch.device("cuda", 0)
    net = _make_net(64 << 10)
    net.set_parameters(p0_a)
    x, m = _data()
    if sync_bn:
        xs, ms = torch.from_numpy(x).to(dev), torch.from_numpy(m).to(dev)
        net.forward_backward_device(xs.data_ptr(), ms.data_ptr(), BATCH)
        net.synchronize()
        want = np.asarray(net.get_gradients(), dtype=np.float64)
        tol = 2e-2     # bf16 activations: the two runs round different partial sums
    else:
        grads = []
        for r in range(2):
            lo, hi = dp.shard_range(BATCH, r, 2)
            xs, ms = torch.from_numpy(x[lo:hi]).to(dev), torch.from_numpy(m[lo:hi]).to(dev)
            net.forward_backward_device(xs.data_ptr(), ms.data_ptr(), hi - lo)
            net.synchronize()
            grads.append(np.asarray(net.get_gradients(), dtype=np.float64))
        want = (grads[0] + grads[1]) / 2
        tol = 1e-6
    scale = np.abs(want).max()
    assert np.abs(g_a - want).max() <= tol * scale + 1e-12, np.abs(g_a - want).max() / scale
