"""The two large benchmark configurations on one GPU, and 64-bit indexing.

  config 4 shape: 4 down / 4 up base-64 graph on 576x576 tiles (the pad-same form of the 572x572 tiling),
                  batch 8 = the per-GPU batch of an 8-GPU run at global batch 64
  config 5 shape: base 128, 1024x1024, batch 16 per GPU: the full-resolution tensors hold 2^31
                  elements (the reference's int element counts, cuda_lib.h:29-270, cannot address them)

A float64 oracle of these steps takes minutes to hours of CPU, so the full-size checks are
size-independent properties (bit-identical repeats, per-sample independence of the forward pass at
element offsets beyond 2^31, agreement of the bf16 and fp32 paths on the loss) plus one image of the
config-4 shape against the bf16-emulating float64 oracle.
"""
import os

import numpy as np
import pytest
import torch

import torch_ref
from util import assert_close_bf16_net

pytestmark = pytest.mark.gpu
os.environ.setdefault("UNET_QUIET", "1")
DEV = "cuda:0"


@pytest.fixture(scope="module")
def nn():
    import neural_net
    return neural_net


@pytest.fixture(scope="module")
def capi():
    from unet_amd import capi as c
    c.lib()
    return c


def _make(nn, graph, dtype, seed=11):
    old = os.environ.get("UNET_DTYPE")
    os.environ["UNET_DTYPE"] = dtype
    try:
        net = nn.NeuralNet()
        net.set_seed(seed)
        net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
    finally:
        if old is None:
            del os.environ["UNET_DTYPE"]
        else:
            os.environ["UNET_DTYPE"] = old
    return net


def p(t):
    return t.data_ptr()


# ---- 64-bit indexing at the operator level -----------------------------------------------------------
# 17 images of 1024 x 1024 x 128 bf16: image 16 starts at element 2^31.  Each operator runs on the whole
# tensor and on the last image alone (a pointer to it, batch 1); the results for that image must be
# bit-identical, which they are only if no offset was truncated to 32 bits on the way.
BIG_N, BIG_HW, BIG_C = 17, 1024, 128


@pytest.fixture(scope="module")
def big():
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    x = torch.empty((BIG_N, BIG_HW, BIG_HW, BIG_C), dtype=torch.bfloat16, device=DEV)
    for i in range(BIG_N):      # filled per image: a float32 temporary of the whole tensor would be 9 GB
        x[i] = torch.randn((BIG_HW, BIG_HW, BIG_C), generator=g, device=DEV, dtype=torch.float32).bfloat16()
    assert x.numel() > 2 ** 31 and (BIG_N - 1) * BIG_HW * BIG_HW * BIG_C >= 2 ** 31
    yield x
    del x
    torch.cuda.empty_cache()


def test_conv3x3_beyond_2_31_elements(capi, big):
    n, hw, c = BIG_N, BIG_HW, BIG_C
    co = 64
    g = torch.Generator(device=DEV)
    g.manual_seed(6)
    w = torch.randn(c * co * 9, generator=g, device=DEV) * (1.0 / np.sqrt(9 * c))
    bias = torch.randn(co, generator=g, device=DEV)
    fwd = torch.zeros(9 * co * c, dtype=torch.bfloat16, device=DEV)
    capi.unet_pack_weights_bf16(p(w), p(fwd), None, c, co, c, co, 3, None)
    rows, rows1 = capi.lib().unet_conv3x3_stat_rows(n, hw, hw, c, co), capi.lib().unet_conv3x3_stat_rows(1, hw, hw, c, co)
    out = torch.full((n, hw, hw, co), float("nan"), dtype=torch.bfloat16, device=DEV)
    stats = torch.full((rows, co, 2), float("nan"), device=DEV)
    capi.unet_conv3x3_bf16(p(big), c, p(fwd), p(bias), co, p(out), co, p(stats), n, hw, hw, c, co, 0, None)
    out1 = torch.full((1, hw, hw, co), float("nan"), dtype=torch.bfloat16, device=DEV)
    stats1 = torch.full((rows1, co, 2), float("nan"), device=DEV)
    capi.unet_conv3x3_bf16(p(big[n - 1]), c, p(fwd), p(bias), co, p(out1), co, p(stats1), 1, hw, hw, c, co, 0, None)
    torch.cuda.synchronize()
    assert torch.isfinite(out1.float()).all()
    assert torch.equal(out[n - 1].view(torch.int16), out1[0].view(torch.int16))
    # the batch-norm partial sums add up to the sums of the output tensor, for the batch and for the last image
    for st, o in ((stats, out), (stats1, out1)):
        tot = st.double().sum(dim=0)
        want = torch.stack([o[i].double().sum(dim=(0, 1)) for i in range(o.shape[0])]).sum(dim=0)
        want2 = torch.stack([(o[i].double() ** 2).sum(dim=(0, 1)) for i in range(o.shape[0])]).sum(dim=0)
        assert torch.allclose(tot[:, 0], want, rtol=1e-4, atol=1e-2 * float(want2.max()) ** 0.5)
        assert torch.allclose(tot[:, 1], want2, rtol=1e-4)
    # and the first image too (offsets far below 2^31): rules out a wrap-around that lands on image 0
    capi.unet_conv3x3_bf16(p(big), c, p(fwd), p(bias), co, p(out1), co, p(stats1), 1, hw, hw, c, co, 0, None)
    torch.cuda.synchronize()
    assert torch.equal(out[0].view(torch.int16), out1[0].view(torch.int16))
    assert not torch.equal(out[0].view(torch.int16), out[n - 1].view(torch.int16))


def test_bn_act_pool_beyond_2_31_elements(capi, big):
    n, hw, c = BIG_N, BIG_HW, BIG_C
    nf = capi.lib().unet_bn_coef_floats(c)
    g = torch.Generator(device=DEV)
    g.manual_seed(7)
    gb = torch.randn(2 * c, generator=g, device=DEV)
    running = torch.rand(2 * c, generator=g, device=DEV) + 0.5
    coef = torch.zeros(nf, device=DEV)
    capi.unet_bn_inference_coef_bf16(p(gb), p(running), c, c, p(coef), None)

    def run(x, batch):
        y = torch.full((batch, hw, hw, c), float("nan"), dtype=torch.bfloat16, device=DEV)
        pooled = torch.full((batch, hw // 2, hw // 2, c), float("nan"), dtype=torch.bfloat16, device=DEV)
        mask = torch.full((batch, hw // 2, hw // 2, c), 255, dtype=torch.uint8, device=DEV)
        capi.unet_bn_act_pool_bf16(p(x), c, p(y), c, p(coef), p(pooled), c, p(mask), batch, hw, hw, c, 0, None)
        y2 = torch.full((batch, hw, hw, c), float("nan"), dtype=torch.bfloat16, device=DEV)
        capi.unet_bn_act_bf16(p(x), c, p(y2), c, p(coef), c, batch * hw * hw, 0, None)
        torch.cuda.synchronize()
        return y, pooled, mask, y2

    y, pooled, mask, y2 = run(big, n)
    last = tuple(t[n - 1].clone() for t in (y, pooled, mask, y2))
    del y, pooled, mask, y2
    y, pooled, mask, y2 = run(big[n - 1], 1)
    for a, b in zip(last, (y, pooled, mask, y2)):
        if a.dtype == torch.bfloat16:
            assert torch.isfinite(b.float()).all()
            a, b = a.view(torch.int16), b[0].view(torch.int16)
        else:
            b = b[0]
        assert torch.equal(a, b)
    # pool backward routes through the tie mask of the last image
    dy = pooled
    dx = torch.full((1, hw, hw, c), float("nan"), dtype=torch.bfloat16, device=DEV)
    capi.unet_max_pool2x2_backward_bf16(p(dy), c, p(mask), p(dx), c, 1, hw, hw, c, 0, None)
    torch.cuda.synchronize()
    assert torch.isfinite(dx.float()).all()


def test_wgrad_beyond_2_31_elements(capi, big):
    """The weight gradient sums over every image: with dz zero everywhere but in the last image the sum must
    equal the gradient of the last image alone (same operands, so only the order of fp32 additions differs)."""
    n, hw, c = BIG_N, BIG_HW, BIG_C
    co = 64
    g = torch.Generator(device=DEV)
    g.manual_seed(8)
    dz = torch.zeros((n, hw, hw, co), dtype=torch.bfloat16, device=DEV)
    dz[n - 1] = torch.randn((hw, hw, co), generator=g, device=DEV).bfloat16()
    gw = torch.full((c * co * 9,), float("nan"), device=DEV)
    gw1 = torch.full((c * co * 9,), float("nan"), device=DEV)
    capi.unet_conv3x3_wgrad_bf16(p(dz), co, p(big), c, p(gw), n, hw, hw, c, co, c, co, None)
    capi.unet_conv3x3_wgrad_bf16(p(dz[n - 1]), co, p(big[n - 1]), c, p(gw1), 1, hw, hw, c, co, c, co, None)
    torch.cuda.synchronize()
    a, b = gw.cpu().numpy().astype(np.float64), gw1.cpu().numpy().astype(np.float64)
    assert np.isfinite(a).all() and np.abs(b).max() > 1.0
    assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max(), np.abs(a - b).max() / np.abs(b).max()


# ---- config 4: 576 x 576 tiles, batch 8 per GPU ---------------------------------------------------------
def test_config4_shape_train_step(nn):
    hw, n = 576, 8
    graph = torch_ref.unet_graph(hw, hw, 3, 64, 4, True)
    rng = np.random.default_rng(41)
    x = rng.standard_normal((n, 3, hw, hw)).astype(np.float32)
    m = (rng.random((n, hw, hw)) < 0.03).astype(np.float32)
    a = _make(nn, graph, "bf16")
    assert a.compute_dtype() == "bf16"
    p0 = a.get_parameters()
    la = a.train_step(x, m, 1e-3, False)
    ga = a.get_gradients()
    lb = a.train_step(x, m, 1e-3, False)       # the same step again on the same network
    assert np.isfinite(la) and np.isfinite(ga).all()
    assert la == lb and np.array_equal(ga, a.get_gradients()), "repeat runs must be bit-identical"
    # per-sample independence of inference
    yb = a.forward(x[:2], True)
    assert np.array_equal(yb[1], a.forward(x[1:2], True)[0])
    # one image against the float64 oracle that rounds to bf16 where the engine stores bf16
    l1 = a.train_step(x[:1], m[:1], 1e-3, False)
    y1 = a.layer_output(len(graph) - 1, 1)
    del a
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        acts = torch_ref.forward(graph, torch_ref.split_params(graph, p0), torch.from_numpy(x[:1]).double(),
                                 training=True, emulate_bf16=True)
        ref_loss = torch_ref.dice_loss(acts[-1], torch.from_numpy(m[:1]).double()).item()
    assert abs(l1 - ref_loss) <= 2e-3 * abs(ref_loss), (l1, ref_loss)
    assert_close_bf16_net(y1, acts[-1].numpy(), "config-4 network output", rel_bar=2e-2, guard=False)
    # the engine's fp32 path on the same batch: the loss agrees to the bf16 bar
    f = _make(nn, graph, "fp32")
    f.set_parameters(p0)
    lf = f.train_step(x, m, 1e-3, False)
    assert abs(la - lf) <= 2e-2 * abs(lf), (la, lf)


# ---- config 5: base 128, 1024 x 1024, batch 16 per GPU ---------------------------------------------------
def test_config5_shape_train_step_and_per_sample_identity(nn):
    hw, n = 1024, 16
    graph = torch_ref.unet_graph(hw, hw, 3, 128, 4, True)
    rng = np.random.default_rng(51)
    x = rng.standard_normal((n, 3, hw, hw)).astype(np.float32)
    m = (rng.random((n, hw, hw)) < 0.03).astype(np.float32)
    free0, _ = torch.cuda.mem_get_info()
    a = _make(nn, graph, "bf16")
    assert a.compute_dtype() == "bf16"
    la = a.train_step(x, m, 1e-3, False)
    free1, _ = torch.cuda.mem_get_info()
    assert (free0 - free1) > 100e9, "the config-5 working set is expected to be above 100 GB"
    ga = a.get_gradients()
    assert np.isfinite(la) and 0.0 < la < 1.0 and np.isfinite(ga).all()
    # every full-resolution tensor of this net is [16,1024,1024,128] = 2^31 elements (and the concat
    # buffers 2^32): the gradients must come out the same on a second run ...
    lb = a.train_step(x, m, 1e-3, False)
    assert la == lb and np.array_equal(ga, a.get_gradients()), "repeat runs must be bit-identical"
    # ... and the last image of the batch, whose activations live at the highest addresses, must give in
    # inference exactly what it gives alone
    yb = a.forward(x, True)
    assert yb.shape == (n, 1, hw, hw) and np.isfinite(yb).all()
    y_last = a.forward(x[n - 1:], True)
    y_first = a.forward(x[:1], True)
    assert np.array_equal(yb[n - 1], y_last[0])
    assert np.array_equal(yb[0], y_first[0])
    assert not np.array_equal(yb[0], yb[n - 1])
    # gradients of a step on the last image alone against the same image repeated at the end of a batch of
    # 2 are covered by per-sample BN statistics differing; the per-layer gradient norms must all be non-zero
    off = 0
    for kind, sh, _, _ in graph:
        if kind in ("conv", "up"):
            nw = sh[2] * sh[3] * sh[4] * sh[4]
            assert np.linalg.norm(ga[off:off + nw].astype(np.float64)) > 0, (kind, sh)
            off += nw + 3 * sh[3]
