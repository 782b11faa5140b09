"""GPU parity of the attention gate (unet_attention_f32 / unet_attention_backward_f32, the reference's
cudaAttention / cudaAttentionBackward, attention.cu:9-499) through the C ABI: against the PyTorch-CPU float64
golden vectors (tests/golden/attention_small.npz) and against the CPU oracle on larger and ragged shapes.
fp32, rel <= 1e-4 (util.assert_close); sums over many cells (weight gradients) are judged against the
tensor's magnitude."""
import numpy as np
import pytest
import torch

from util import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    return gpu_util


@pytest.fixture(scope="module")
def capi():
    from unet_amd import capi as c
    c.lib()
    return c


def run_forward(G, capi, x, g, params, ci):
    n, cx, hx, wx = x.shape
    cg = g.shape[1]
    xd, gd, pd = G.dev(x), G.dev(g), G.dev(params)
    int1, alpha, out = G.empty((n, ci, hx // 2, wx // 2)), G.empty((n, hx // 2, wx // 2)), G.empty(x.shape)
    capi.unet_attention_f32(G.p(int1), G.p(alpha), G.p(out), G.p(pd), G.p(xd), G.p(gd), n, hx, wx, cx, cg, ci, None)
    return (xd, gd, pd), (int1, alpha, out)


def run_backward(G, capi, dy, held, fwd, shape):
    n, cx, cg, ci, hx, wx = shape
    (xd, gd, pd), (int1, alpha, _) = held, fwd
    dyd = G.dev(dy)
    g1, g2 = G.empty(tuple(int1.shape)), G.empty(tuple(alpha.shape))
    dx, dg, dp = G.empty(tuple(xd.shape)), G.empty(tuple(gd.shape)), G.empty(tuple(pd.shape))
    capi.unet_attention_backward_f32(G.p(dyd), G.p(int1), G.p(alpha), G.p(xd), G.p(gd), G.p(pd), G.p(g1), G.p(g2),
                                     G.p(dx), G.p(dg), G.p(dp), n, hx, wx, cx, cg, ci, None)
    return G.host(dx), G.host(dg), G.host(dp), G.host(g1), G.host(g2)


@pytest.mark.parametrize("name", ["attn_a", "attn_odd", "attn_b"])
def test_attention_gate_against_torch_golden(G, capi, golden_attention, name):
    A = golden_attention
    n, cx, cg, ci, hx, wx = (int(v) for v in A[name + "_cfg"])
    assert capi.lib().unet_attention_param_count(cx, cg, ci) == A[name + "_params"].size
    held, fwd = run_forward(G, capi, A[name + "_x"], A[name + "_g"], A[name + "_params"], ci)
    assert_close(G.host(fwd[0]), A[name + "_int1"], what=name + " int1")
    assert_close(G.host(fwd[1]), A[name + "_alpha"], what=name + " alpha")
    assert_close(G.host(fwd[2]), A[name + "_y"], what=name + " y")
    dx, dg, dp, _, _ = run_backward(G, capi, A[name + "_dy"], held, fwd, (n, cx, cg, ci, hx, wx))
    assert_close(dx, A[name + "_dx"], atol_scale=1e-5, what=name + " dx")
    assert_close(dg, A[name + "_dg"], atol_scale=1e-5, what=name + " dg")
    assert_close(dp, A[name + "_dparams"], atol_scale=1e-5, what=name + " dparams")


@pytest.mark.parametrize("shape", [(2, 64, 128, 32, 64, 64),     # the layer's ratio at a real size
                                   (1, 12, 20, 6, 18, 22),       # H/2, W/2 odd; partial 4x4 blocks; ragged channels
                                   (3, 40, 8, 37, 8, 4),         # more intermediate channels than one 32-tile
                                   (1, 3, 5, 1, 2, 2)])          # smallest legal image
def test_attention_gate_against_oracle(G, capi, oracle, shape):
    n, cx, cg, ci, hx, wx = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal((n, cx, hx, wx)).astype(np.float32)
    g = rng.standard_normal((n, cg, hx // 2, wx // 2)).astype(np.float32)
    params = (rng.standard_normal(oracle.attention_params(cx, cg, ci)) * 0.2).astype(np.float32)
    dy = rng.standard_normal(x.shape).astype(np.float32)
    held, fwd = run_forward(G, capi, x, g, params, ci)
    y_ref, int1_ref, alpha_ref = oracle.attention(x, g, params, ci)
    assert_close(G.host(fwd[0]), int1_ref, atol_scale=1e-5, what="int1")
    assert_close(G.host(fwd[1]), alpha_ref, what="alpha")
    assert_close(G.host(fwd[2]), y_ref, what="y")
    # the oracle's backward takes the device's own int1 / alpha so that ReLU decisions at |z| ~ 1e-7 agree
    dx, dg, dp, g1, g2 = run_backward(G, capi, dy, held, fwd, shape)
    dx_ref, dg_ref, dp_ref = oracle.attention_backward(dy, G.host(fwd[0]), G.host(fwd[1]), x, g, params)
    assert_close(dx, dx_ref, atol_scale=1e-5, what="dx")
    assert_close(dg, dg_ref, atol_scale=1e-5, what="dg")
    assert_close(dp, dp_ref, atol_scale=2e-5, what="dparams")
    # twice the same call: bit-identical (no atomics anywhere)
    again = run_backward(G, capi, dy, held, fwd, shape)
    for a, b in zip((dx, dg, dp, g1, g2), again):
        np.testing.assert_array_equal(a, b)


def test_attention_gate_rejects_bad_shapes(G, capi):
    t = G.empty((64,))
    for hx, wx in ((5, 4), (4, 0)):
        with pytest.raises(RuntimeError):
            capi.unet_attention_f32(G.p(t), G.p(t), G.p(t), G.p(t), G.p(t), G.p(t), 1, hx, wx, 1, 1, 1, None)


def _bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).bfloat16().float().numpy()


def _nhwc(G, a, ld):
    n, c, h, w = a.shape
    t = torch.zeros((n, h, w, ld), dtype=torch.bfloat16, device=G.DEV)
    t[..., :c] = torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 3, 1))).to(G.DEV).bfloat16()
    return t


def _nchw(t, c):
    return t[..., :c].float().cpu().numpy().transpose(0, 3, 1, 2)


@pytest.mark.parametrize("shape", [(2, 64, 128, 32, 32, 48), (1, 32, 64, 16, 18, 22), (2, 96, 32, 48, 8, 8)])
def test_attention_gate_between_bf16_tensors(G, capi, oracle, shape):
    """unet_attention_bf16 / unet_attention_backward_bf16: NHWC bf16 views (strides wider than the channel count),
    two incoming gradients, written and accumulated input gradients -- against the oracle on the same bf16-rounded
    operands (bf16 outputs: rel <= 2e-2; fp32 parameter gradients against their magnitude)."""
    from util import assert_close_bf16
    n, cx, cg, ci, hx, wx = shape
    rng = np.random.default_rng(sum(shape) + 1)
    x = _bf16(rng.standard_normal((n, cx, hx, wx)))
    g = _bf16(rng.standard_normal((n, cg, hx // 2, wx // 2)))
    params = (rng.standard_normal(oracle.attention_params(cx, cg, ci)) * 0.1).astype(np.float32)
    dy1 = _bf16(rng.standard_normal(x.shape))
    dy2 = _bf16(rng.standard_normal(x.shape))
    dx0 = _bf16(rng.standard_normal(x.shape))          # what the x gradient buffer already holds
    xt, gt = _nhwc(G, x, cx + 16), _nhwc(G, g, cg + 8)
    pd = G.dev(params)
    int1, alpha = G.empty((n, ci, hx // 2, wx // 2)), G.empty((n, hx // 2, wx // 2))
    out = torch.full((n, hx, wx, cx + 32), 7.0, dtype=torch.bfloat16, device=G.DEV)
    capi.unet_attention_bf16(G.p(xt), cx + 16, G.p(gt), cg + 8, G.p(pd), G.p(int1), G.p(alpha), out.data_ptr() + 32,
                             cx + 32, n, hx, wx, cx, cg, ci, None)
    y_ref, int1_ref, alpha_ref = oracle.attention(x, g, params, ci)
    oh = out.float().cpu().numpy()
    assert np.all(oh[..., :16] == 7.0) and np.all(oh[..., 16 + cx:] == 7.0)   # only the slice was written
    assert_close(G.host(int1), int1_ref, atol_scale=1e-5, what="int1")
    assert_close(G.host(alpha), alpha_ref, what="alpha")
    assert_close_bf16(oh[..., 16:16 + cx].transpose(0, 3, 1, 2), y_ref, what="gated output")

    d1, d2 = _nhwc(G, dy1, cx), _nhwc(G, dy2, cx + 8)
    dx, dg = _nhwc(G, dx0, cx + 8), torch.full((n, hx // 2, wx // 2, cg), float("nan"), dtype=torch.bfloat16, device=G.DEV)
    dp = G.empty((params.size,))
    capi.unet_attention_backward_bf16(G.p(d1), cx, G.p(d2), cx + 8, G.p(xt), cx + 16, G.p(gt), cg + 8, G.p(pd), G.p(int1),
                                      G.p(alpha), G.p(dx), cx + 8, 1, G.p(dg), cg, 0, G.p(dp), n, hx, wx, cx, cg, ci, None)
    dx_ref, dg_ref, dp_ref = oracle.attention_backward(dy1 + dy2, G.host(int1), G.host(alpha), x, g, params)
    assert_close_bf16(_nchw(dx, cx), dx_ref + dx0, what="dx (accumulated)")
    assert_close_bf16(_nchw(dg, cg), dg_ref, what="dg")
    assert np.abs(G.host(dp) - dp_ref).max() <= 1e-4 * np.abs(dp_ref).max() + 1e-5
    # one gradient source, written dx, accumulated dg
    dx2 = torch.full((n, hx, wx, cx), float("nan"), dtype=torch.bfloat16, device=G.DEV)
    capi.unet_attention_backward_bf16(G.p(d1), cx, None, 0, G.p(xt), cx + 16, G.p(gt), cg + 8, G.p(pd), G.p(int1),
                                      G.p(alpha), G.p(dx2), cx, 0, G.p(dg), cg, 1, G.p(dp), n, hx, wx, cx, cg, ci, None)
    dx1_ref, dg1_ref, _ = oracle.attention_backward(dy1, G.host(int1), G.host(alpha), x, g, params)
    assert_close_bf16(_nchw(dx2, cx), dx1_ref, what="dx (written)")
    assert_close_bf16(_nchw(dg, cg), _bf16(dg_ref) + dg1_ref, what="dg (accumulated)")
