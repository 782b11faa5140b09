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
