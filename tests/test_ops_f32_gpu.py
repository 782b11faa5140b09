"""GPU parity tests of the fp32 operator set: HIP kernels (through the C ABI)
vs the CPU oracle and the committed golden vectors.

Bar: fp32 relative <= 1e-4 (see util.assert_close); pool values / indices /
tie masks, concat layout and shapes bit-exact.
"""
import ctypes

import numpy as np
import pytest
import torch

from util import assert_close, assert_exact

pytestmark = pytest.mark.gpu

CONV = ["conv_a", "conv_b", "conv_c", "conv_d", "conv_e", "conv_f", "conv_g"]


@pytest.fixture(scope="module")
def G():
    import gpu_util
    assert torch.cuda.is_available(), "these tests need the GPU"
    return gpu_util


@pytest.fixture(scope="module")
def capi():
    from unet_amd import capi as c
    c.lib()
    return c


# ------------------------------------------------------------------ conv ----
@pytest.mark.parametrize("name", CONV)
def test_conv_golden(G, golden, name):
    n, ci, co, h, w, k, pad, s = (int(v) for v in golden[name + "_cfg"])
    x, wt, dy = G.dev(golden[name + "_x"]), G.dev(golden[name + "_w"]), G.dev(golden[name + "_dy"])
    y = G.convolve(x, wt, co, k, pad, s)
    assert_close(G.host(y), golden[name + "_y"], what=name + " y")
    dx, dw = G.convolve_backward(dy, x, wt, k, pad, s)
    assert_close(G.host(dx), golden[name + "_dx"], what=name + " dx")
    assert_close(G.host(dw), golden[name + "_dw"], what=name + " dw")


@pytest.mark.parametrize("cfg", [
    # n, ci, co, h, w, k, pad, stride
    (2, 3, 64, 40, 40, 3, 1, 1),     # the reference test's channel counts, smaller image
    (1, 20, 18, 33, 17, 3, 1, 1),    # ragged tiles, channel counts off the block sizes
    (2, 64, 64, 40, 40, 3, 1, 1),    # matrix-core kernels forward and backward: 32-wide tiles, ragged at the border
    (1, 130, 70, 16, 16, 3, 1, 1),   # several channel blocks, 16-wide tiles, the data gradient split over input channels
    (3, 32, 48, 64, 33, 3, 1, 1),    # odd width, many 64-pixel regions per workgroup
    (2, 5, 7, 21, 30, 3, 1, 2),
    (1, 4, 6, 19, 19, 5, 2, 1),
    (1, 3, 4, 15, 18, 7, 3, 1),
    (3, 9, 17, 12, 12, 1, 0, 1),
    (1, 2, 3, 11, 11, 3, 0, 2),
    (1, 1, 1, 1, 1, 1, 0, 1),        # smallest possible
])
def test_conv_vs_oracle(G, oracle, cfg):
    n, ci, co, h, w, k, pad, s = cfg
    rng = np.random.default_rng(hash(cfg) % (2**32))
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal(ci * co * k * k + co) * 0.2).astype(np.float32)
    y_ref = oracle.convolve(x, wt, co, k, pad, s)
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref, dw_ref = oracle.convolve_backward(x, dy, wt, k, pad, s)
    xd, wd, dyd = G.dev(x), G.dev(wt), G.dev(dy)
    assert_close(G.host(G.convolve(xd, wd, co, k, pad, s)), y_ref, what=f"conv {cfg}")
    dx, dw = G.convolve_backward(dyd, xd, wd, k, pad, s)
    assert_close(G.host(dx), dx_ref, what=f"conv dx {cfg}")
    assert_close(G.host(dw), dw_ref, what=f"conv dw {cfg}")
    # skipping the data gradient must not change the weight gradient
    _, dw2 = G.convolve_backward(dyd, xd, wd, k, pad, s, want_dx=False)
    assert_exact(G.host(dw2), G.host(dw), what="dw without dx")


def test_conv_linearity_full_size(G):
    """Reference test shape (convolution_test.h:157-165): N16, 3->64, 256x256.
    Too slow for the scalar oracle, so checked through linearity:
    conv(a*x1 + x2) == a*conv(x1) + conv(x2) - (a)*bias-term."""
    n, ci, co, h, k = 16, 3, 64, 256, 3
    g = torch.Generator(device="cpu").manual_seed(1)
    x1 = torch.randn(n, ci, h, h, generator=g).to(G.DEV)
    x2 = torch.randn(n, ci, h, h, generator=g).to(G.DEV)
    wt = (torch.randn(ci * co * k * k + co, generator=g) * 0.2).to(G.DEV)
    wt[-co:] = 0  # zero bias makes the op linear
    y1, y2 = G.convolve(x1, wt, co, k, 1, 1), G.convolve(x2, wt, co, k, 1, 1)
    y3 = G.convolve(2.0 * x1 + x2, wt, co, k, 1, 1)
    assert_close(G.host(y3), G.host(2.0 * y1 + y2), rtol=1e-4, atol_scale=2e-6, what="linearity")
    # spot-check 64 output pixels against a float64 evaluation of the definition
    xh, wh = x1.cpu().double().numpy(), wt.cpu().double().numpy()
    yh = G.host(y1)
    rng = np.random.default_rng(0)
    W4 = wh[:ci * co * 9].reshape(ci, co, 3, 3)
    for _ in range(64):
        b, c, i, j = rng.integers(n), rng.integers(co), rng.integers(h), rng.integers(h)
        acc = 0.0
        for kh in range(3):
            for kw in range(3):
                ii, jj = i - 1 + kh, j - 1 + kw
                if 0 <= ii < h and 0 <= jj < h:
                    acc += float(np.dot(xh[b, :, ii, jj], W4[:, c, kh, kw]))
        assert abs(acc - yh[b, c, i, j]) <= 1e-4 * max(abs(acc), abs(yh[b, c, i, j])) + 1e-5


def test_conv_rejects_bad_arguments(G, capi):
    x, w, y = G.empty((1, 1, 4, 4)), G.empty((10,)), G.empty((1, 1, 4, 4))
    with pytest.raises(RuntimeError, match="square and odd"):
        capi.unet_convolve_f32(G.p(w), G.p(x), G.p(y), 16, 1, 2, 1, 1, 4, 4, 4, 4, 1, 1, None)
    with pytest.raises(RuntimeError, match="do not match"):
        capi.unet_convolve_f32(G.p(w), G.p(x), G.p(y), 16, 1, 3, 1, 1, 5, 4, 4, 4, 1, 1, None)
    with pytest.raises(RuntimeError, match="NULL"):
        capi.unet_convolve_f32(None, G.p(x), G.p(y), 16, 1, 3, 1, 1, 4, 4, 4, 4, 1, 1, None)


# ------------------------------------------------------------ batch norm ----
def _bn_run(G, capi, x, gb, run_in, dy):
    n, c, h, w = x.shape
    xd, gbd, dyd = G.dev(x), G.dev(gb), G.dev(dy)
    y, stats, run = G.empty(x.shape), G.empty((2 * c,)), G.dev(run_in)
    capi.unet_batch_norm_train_f32(G.p(xd), G.p(y), G.p(gbd), G.p(stats), G.p(run), c, h, w, n, None)
    dx, dgb = G.empty(x.shape), G.empty((2 * c,))
    capi.unet_batch_norm_backward_f32(G.p(dyd), G.p(xd), G.p(stats), G.p(gbd), G.p(dx), G.p(dgb), c, h, w, n, None)
    yi = G.empty(x.shape)
    rund = G.dev(run_in)
    capi.unet_batch_norm_inference_f32(G.p(xd), G.p(yi), G.p(gbd), G.p(rund), h, w, c, n, None)
    return [G.host(t) for t in (y, stats, run, dx, dgb, yi)]


@pytest.mark.parametrize("name", ["bn_a", "bn_b"])
def test_batch_norm_golden(G, capi, golden, name):
    y, stats, run, dx, dgb, yi = _bn_run(G, capi, golden[name + "_x"], golden[name + "_gb"],
                                         golden[name + "_run_in"], golden[name + "_dy"])
    assert_close(y, golden[name + "_y"], what=name + " y")
    assert_close(stats, golden[name + "_stats"], what=name + " raw sums")
    assert_close(run, golden[name + "_run_out"], what=name + " running")
    assert_close(dx, golden[name + "_dx"], rtol=2e-4, what=name + " dx")
    assert_close(dgb, golden[name + "_dgb"], what=name + " dgamma/dbeta")
    assert_close(yi, golden[name + "_yinf"], what=name + " inference")


@pytest.mark.parametrize("shape", [(4, 64, 32, 32), (3, 5, 17, 9), (1, 1, 1, 1)])
def test_batch_norm_inference_with_relu_equals_the_two_passes(G, capi, shape):
    """unet_batch_norm_inference_relu_f32 = unet_batch_norm_inference_f32 then unet_relu_f32, bit for bit."""
    n, c, h, w = shape
    rng = np.random.default_rng(c)
    x = G.dev((rng.standard_normal(shape) * 1.5 + 0.2).astype(np.float32))
    gb = G.dev(rng.standard_normal(2 * c).astype(np.float32))
    run = np.empty(2 * c, np.float32)
    run[0::2], run[1::2] = rng.standard_normal(c), rng.random(c) + 0.1
    run = G.dev(run)
    y, two, one = G.empty(shape), G.empty(shape), G.empty(shape)
    capi.unet_batch_norm_inference_f32(G.p(x), G.p(y), G.p(gb), G.p(run), h, w, c, n, None)
    capi.unet_relu_f32(G.p(y), G.p(two), n * c * h * w, None)
    capi.unet_batch_norm_inference_relu_f32(G.p(x), G.p(one), G.p(gb), G.p(run), h, w, c, n, None)
    got, want = G.host(one), G.host(two)
    np.testing.assert_array_equal(got, want)
    assert (want == 0).any() or n * c * h * w == 1
    assert (want > 0).any() or n * c * h * w == 1


@pytest.mark.parametrize("shape", [(16, 64, 64, 64), (3, 5, 17, 9), (1, 1, 1, 1), (2, 130, 8, 8)])
def test_batch_norm_vs_oracle(G, capi, oracle, shape):
    n, c, h, w = shape
    rng = np.random.default_rng(n * 1000 + c)
    x = (rng.standard_normal(shape) * 1.5 + 0.7).astype(np.float32)
    gb = rng.standard_normal(2 * c).astype(np.float32)
    run_in = np.abs(rng.standard_normal(2 * c)).astype(np.float32)
    dy = rng.standard_normal(shape).astype(np.float32)
    y, stats, run, dx, dgb, yi = _bn_run(G, capi, x, gb, run_in, dy)
    y_ref, stats_ref, run_ref = oracle.bn_train(x, gb, run_in)
    dx_ref, dgb_ref = oracle.bn_backward(x, dy, gb, stats_ref)
    if n * h * w > 1:
        assert_close(y, y_ref, what="bn y")
        assert_close(dx, dx_ref, rtol=2e-4, atol_scale=1e-5, what="bn dx")
    assert_close(stats, stats_ref, what="bn raw sums")
    assert_close(run, run_ref, what="bn running stats")
    assert_close(dgb, dgb_ref, atol_scale=2e-6, what="bn dgamma/dbeta")
    assert_close(yi, oracle.bn_inference(x, gb, run_in), what="bn inference")


# -------------------------------------------------------------- max pool ----
def _pool_run(G, capi, x, dy, k=2, s=2):
    n, c, h, w = x.shape
    ho, wo = h // s, w // s
    xd, dyd = G.dev(x), G.dev(dy)
    y, idx, tm = G.empty((n, c, ho, wo)), G.empty((n, c, ho, wo), torch.int32), G.empty((n, c, ho, wo), torch.uint8)
    capi.unet_max_pool_indices_f32(G.p(xd), G.p(y), G.p(idx), G.p(tm), n, s, k, c, h, w, None)
    y2 = G.empty((n, c, ho, wo))
    capi.unet_max_pool_f32(G.p(xd), G.p(y2), n, s, k, c, h, w, None)
    dx_t, dx_i = G.empty(x.shape), G.empty(x.shape)
    capi.unet_max_pool_backward_f32(G.p(dyd), G.p(xd), G.p(y), G.p(dx_t), n, s, c, h, w, None)
    capi.unet_max_pool_backward_indices_f32(G.p(dyd), G.p(idx), G.p(dx_i), n, c, h, w, ho, wo, None)
    return [G.host(t) for t in (y, idx, tm, y2, dx_t, dx_i)]


@pytest.mark.parametrize("name", ["pool_a", "pool_b"])
def test_max_pool_golden(G, capi, golden, name):
    y, idx, tm, y2, dx_t, dx_i = _pool_run(G, capi, golden[name + "_x"], golden[name + "_dy"])
    assert_exact(y, golden[name + "_y"], what=name + " y")
    assert_exact(y2, golden[name + "_y"], what=name + " y (no indices)")
    assert_exact(idx, golden[name + "_idx"], what=name + " indices")
    assert_exact(tm, golden[name + "_tiemask"], what=name + " tie mask")
    assert_exact(dx_t, golden[name + "_dx_ties"], what=name + " dx ties")
    assert_exact(dx_i, golden[name + "_dx_first"], what=name + " dx indices")


@pytest.mark.parametrize("shape,ties", [((8, 64, 64, 64), False), ((2, 3, 9, 13), True), ((1, 1, 2, 2), True),
                                        ((2, 1030, 4, 6), True)])
def test_max_pool_vs_oracle(G, capi, oracle, shape, ties):
    rng = np.random.default_rng(shape[1])
    x = rng.standard_normal(shape).astype(np.float32)
    if ties:
        x = np.maximum(np.round(x * 2) / 2, 0).astype(np.float32)
    ho, wo = shape[2] // 2, shape[3] // 2
    dy = rng.standard_normal(shape[:2] + (ho, wo)).astype(np.float32)
    y, idx, tm, y2, dx_t, dx_i = _pool_run(G, capi, x, dy)
    y_ref, idx_ref, tm_ref = oracle.maxpool(x, 2, 2, want_indices=True, want_tiemask=True)
    assert_exact(y, y_ref, what="pool y")
    assert_exact(y2, y_ref, what="pool y (no indices)")
    assert_exact(idx, idx_ref, what="pool indices")
    assert_exact(tm, tm_ref, what="pool tie mask")
    assert_exact(dx_t, oracle.maxpool_backward(dy, x, y_ref, 2), what="pool dx ties")
    assert_exact(dx_i, oracle.maxpool_backward_indices(dy, idx_ref, shape[2], shape[3]), what="pool dx indices")


# -------------------------------------------------------------- upsample ----
def _up_run(G, capi, x, wt, dy, co, s):
    n, ci, h, w = x.shape
    xd, wd, dyd = G.dev(x), G.dev(wt), G.dev(dy)
    y = G.empty((n, co, h * s, w * s))
    capi.unet_upsample_f32(G.p(xd), G.p(wd), G.p(y), s, ci, co, h, w, n, None)
    dx, dw = G.empty(x.shape), G.empty(wt.shape)
    capi.unet_upsample_backward_f32(G.p(dyd), G.p(xd), G.p(wd), G.p(dx), G.p(dw), s, ci, co, h, w, n, None)
    return [G.host(t) for t in (y, dx, dw)]


@pytest.mark.parametrize("name", ["up_a", "up_b", "up_c"])
def test_upsample_golden(G, capi, golden, name):
    n, ci, co, h, w, s = (int(v) for v in golden[name + "_cfg"])
    y, dx, dw = _up_run(G, capi, golden[name + "_x"], golden[name + "_w"], golden[name + "_dy"], co, s)
    assert_close(y, golden[name + "_y"], what=name + " y")
    assert_close(dx, golden[name + "_dx"], what=name + " dx")
    assert_close(dw, golden[name + "_dw"], what=name + " dw")


@pytest.mark.parametrize("cfg", [(4, 64, 32, 32, 32, 2), (1, 19, 9, 7, 5, 2), (2, 3, 5, 4, 4, 3), (1, 2, 2, 3, 2, 4),
                                 # the matrix-core forward and weight gradient: 16-wide regions, several channel blocks; ragged
                                 # regions and channel counts; odd width with 32-wide regions
                                 (2, 80, 144, 16, 16, 2), (1, 20, 17, 9, 11, 2), (3, 32, 16, 6, 45, 2),
                                 (1, 256, 64, 8, 8, 2),    # forward: input channels split over workgroups
                                 # enough workgroups for the matrix-core data gradient: 32- and 16-wide tiles, ragged
                                 (8, 64, 32, 32, 32, 2), (4, 70, 20, 30, 35, 2), (16, 130, 48, 16, 16, 2)])
def test_upsample_vs_oracle(G, capi, oracle, cfg):
    n, ci, co, h, w, s = cfg
    rng = np.random.default_rng(ci * 100 + co)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal(ci * co * s * s + co) * 0.3).astype(np.float32)
    dy = rng.standard_normal((n, co, h * s, w * s)).astype(np.float32)
    y, dx, dw = _up_run(G, capi, x, wt, dy, co, s)
    assert_close(y, oracle.upsample(x, wt, co, s), what=f"upsample {cfg}")
    dx_ref, dw_ref = oracle.upsample_backward(x, wt, dy, s)
    assert_close(dx, dx_ref, what=f"upsample dx {cfg}")
    assert_close(dw, dw_ref, what=f"upsample dw {cfg}")


# ------------------------------------------------- concat / activations ----
def test_concat_layout(G, capi, oracle, golden):
    a, b = golden["cat_a"], golden["cat_b"]
    n, c1, h, w = a.shape
    c2 = b.shape[1]
    out = G.empty((n, c1 + c2, h, w))
    ad, bd = G.dev(a), G.dev(b)
    capi.unet_concat_f32(G.p(ad), G.p(bd), G.p(out), n, h * w, c1, c2, None)
    assert_exact(G.host(out), golden["cat_y"], what="concat")
    back = G.empty((n * (c1 + c2) * h * w,))
    capi.unet_concat_backward_f32(G.p(out), G.p(back), h, w, c1, c2, n, None)
    back = G.host(back)
    assert_exact(back[:a.size].reshape(a.shape), a, what="concat backward A")
    assert_exact(back[a.size:].reshape(b.shape), b, what="concat backward B")


@pytest.mark.parametrize("n", [1, 3, 1024, 4099, 1 << 20])
def test_elementwise(G, capi, oracle, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32)
    dy = rng.standard_normal(n).astype(np.float32)
    xd, dyd, out = G.dev(x), G.dev(dy), G.empty((n,))
    capi.unet_relu_f32(G.p(xd), G.p(out), n, None)
    assert_exact(G.host(out), oracle.relu(x), what="relu")
    capi.unet_relu_backward_f32(G.p(dyd), G.p(xd), G.p(out), n, None)
    assert_exact(G.host(out), oracle.relu_backward(dy, x), what="relu backward")
    capi.unet_sigmoid_f32(G.p(xd), G.p(out), n, None)
    y = G.host(out).copy()
    assert_close(y, oracle.sigmoid(x), what="sigmoid")
    capi.unet_sigmoid_backward_f32(G.p(dyd), G.p(out), G.p(xd), n, None)
    assert_close(G.host(xd), oracle.sigmoid_backward(dy, y), what="sigmoid backward")
    a = G.dev(x)
    capi.unet_add_b_to_a_f32(G.p(a), G.p(dyd), n, None)
    assert_exact(G.host(a), oracle.add_b_to_a(x, dy), what="add b to a")
    a = G.dev(x)
    capi.unet_threshold_f32(G.p(a), 0.5, n, None)
    assert_exact(G.host(a), oracle.threshold(x, 0.5), what="threshold")
    a = G.dev(x)
    capi.unet_update_weights_sgd_f32(G.p(a), G.p(dyd), 0.1, n, None)
    assert_close(G.host(a), oracle.sgd(x, dy, 0.1), rtol=1e-6, what="sgd")
    # unaligned views take the scalar path
    if n > 8:
        big = G.dev(np.concatenate([[0.0], x]).astype(np.float32))
        o2 = G.empty((n + 1,))
        capi.unet_relu_f32(big.data_ptr() + 4, o2.data_ptr() + 4, n, None)
        assert_exact(G.host(o2)[1:], oracle.relu(x), what="relu unaligned")


# ---------------------------------------------------------------- losses ----
def test_dice(G, capi, oracle, golden):
    mask, pred = golden["dice_mask"], golden["dice_pred"]
    n, h, w = mask.shape
    lv = G.empty((3 * n,))
    d = ctypes.c_float()
    md, pd = G.dev(mask), G.dev(pred)
    capi.unet_dice_score_f32(G.p(md), G.p(pd), h, w, n, G.p(lv), ctypes.byref(d), None)
    assert_close(G.host(lv), golden["dice_lv"], what="dice sums")
    assert_close(np.float32(d.value), golden["dice_score"], what="dice score")
    g = G.empty(mask.shape)
    capi.unet_dice_loss_backward_f32(G.p(md), G.p(pd), G.p(lv), G.p(g), h, w, n, 1.0, None)
    assert_close(G.host(g), golden["dice_grad_ref"], what="dice backward")
    capi.unet_dice_loss_backward_f32(G.p(md), G.p(pd), G.p(lv), G.p(g), h, w, n, -1.0 / n, None)
    assert_close(G.host(g), -golden["dice_grad_ref"] / n, what="dice backward scaled")
    # larger, vs oracle
    rng = np.random.default_rng(5)
    mask = (rng.random((16, 256, 256)) < 0.03).astype(np.float32)
    pred = rng.random((16, 256, 256)).astype(np.float32)
    lv = G.empty((48,))
    md, pd = G.dev(mask), G.dev(pred)
    capi.unet_dice_score_f32(G.p(md), G.p(pd), 256, 256, 16, G.p(lv), ctypes.byref(d), None)
    d_ref, lv_ref = oracle.dice_score(mask, pred)
    assert_close(G.host(lv), lv_ref, what="dice sums 256x256")
    assert_close(np.float32(d.value), np.float32(d_ref), what="dice 256x256")


def test_difference_and_xent(G, capi, oracle, golden):
    y, m = golden["diff_y"], golden["diff_m"]
    g = G.empty(y.shape)
    md, yd = G.dev(m), G.dev(y)
    capi.unet_difference_loss_backward_f32(G.p(md), G.p(yd), G.p(g), y.size, None)
    assert_close(G.host(g), golden["diff_grad"], what="difference loss backward")
    logits, tgt = golden["xent_logits"], golden["xent_target"]
    n, c, h, w = logits.shape
    gl, loss = G.empty(logits.shape), G.empty((1,))
    ld, td = G.dev(logits), G.dev(tgt)
    capi.unet_softmax_xent_f32(G.p(ld), G.p(td), G.p(gl), G.p(loss), n, c, h, w, None)
    assert_close(G.host(loss)[0], golden["xent_loss"], what="cross entropy")
    assert_close(G.host(gl), golden["xent_grad"], what="cross entropy grad")
    rng = np.random.default_rng(9)
    logits = (rng.standard_normal((3, 7, 33, 20)) * 3).astype(np.float32)
    tgt = rng.integers(0, 7, (3, 33, 20)).astype(np.int32)
    gl = G.empty(logits.shape)
    ld, td = G.dev(logits), G.dev(tgt)
    capi.unet_softmax_xent_f32(G.p(ld), G.p(td), G.p(gl), G.p(loss), 3, 7, 33, 20, None)
    l_ref, g_ref = oracle.softmax_xent(logits, tgt)
    assert_close(G.host(loss)[0], np.float32(l_ref), what="cross entropy vs oracle")
    assert_close(G.host(gl), g_ref, what="cross entropy grad vs oracle")


def test_batch_norm_fused_with_relu_is_bit_identical_to_the_two_calls(G, capi):
    """unet_batch_norm_train_relu_f32 = cudaBatchNormTrain + cudaRelu, unet_batch_norm_relu_backward_f32 =
    cudaReluBackward + cudaBatchNormBackward: the same bits, one tensor pass fewer each way."""
    rng = np.random.default_rng(31)
    n, c, h, w = 3, 5, 17, 23
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    gb = (rng.standard_normal(2 * c) * 0.5 + np.tile([1.0, 0.0], c)).astype(np.float32)
    dy = rng.standard_normal((n, c, h, w)).astype(np.float32)
    xd, gbd, dyd = G.dev(x), G.dev(gb), G.dev(dy)
    bn, y2, y1 = G.empty(x.shape), G.empty(x.shape), G.empty(x.shape)
    st2, st1 = G.empty((2 * c,)), G.empty((2 * c,))
    run2, run1 = G.dev(np.tile([0.0, 1.0], c).astype(np.float32)), G.dev(np.tile([0.0, 1.0], c).astype(np.float32))
    capi.unet_batch_norm_train_f32(G.p(xd), G.p(bn), G.p(gbd), G.p(st2), G.p(run2), c, h, w, n, None)
    capi.unet_relu_f32(G.p(bn), G.p(y2), x.size, None)
    capi.unet_batch_norm_train_relu_f32(G.p(xd), G.p(y1), G.p(gbd), G.p(st1), G.p(run1), c, h, w, n, None)
    assert_exact(G.host(y1), G.host(y2))
    assert_exact(G.host(st1), G.host(st2))
    assert_exact(G.host(run1), G.host(run2))
    g2, dx2, dx1 = G.empty(x.shape), G.empty(x.shape), G.empty(x.shape)
    gg2, gg1 = G.empty((2 * c,)), G.empty((2 * c,))
    capi.unet_relu_backward_f32(G.p(dyd), G.p(bn), G.p(g2), x.size, None)
    capi.unet_batch_norm_backward_f32(G.p(g2), G.p(xd), G.p(st2), G.p(gbd), G.p(dx2), G.p(gg2), c, h, w, n, None)
    capi.unet_batch_norm_relu_backward_f32(G.p(dyd), G.p(y1), G.p(xd), G.p(st1), G.p(gbd), G.p(dx1), G.p(gg1), c, h, w, n,
                                           None)
    assert_exact(G.host(dx1), G.host(dx2))
    assert_exact(G.host(gg1), G.host(gg2))


def test_cross_entropy_ignores_labels_that_are_no_class(G, capi, oracle):
    """Ignore labels (255, -1, anything outside [0, C)) must not index the logits: such pixels add no loss, get
    a zero gradient and leave the mean (torch's ignore_index, for every value that is not a class)."""
    rng = np.random.default_rng(10)
    n, c, h, w = 2, 5, 19, 23
    logits = (rng.standard_normal((n, c, h, w)) * 2).astype(np.float32)
    tgt = rng.integers(0, c, (n, h, w)).astype(np.int32)
    tgt[rng.random(tgt.shape) < 0.2] = 255
    tgt[rng.random(tgt.shape) < 0.1] = -1
    tgt[0, 0, 0], tgt[1, 3, 4] = 2 ** 31 - 1, -2 ** 31
    gl, loss = G.empty(logits.shape), G.empty((1,))
    ld, td = G.dev(logits), G.dev(tgt)
    capi.unet_softmax_xent_f32(G.p(ld), G.p(td), G.p(gl), G.p(loss), n, c, h, w, None)
    l_ref, g_ref = oracle.softmax_xent(logits, tgt)
    assert_close(G.host(loss)[0], np.float32(l_ref), what="cross entropy with ignored pixels")
    got = G.host(gl)
    assert_close(got, g_ref, what="cross entropy grad with ignored pixels")
    bad = (tgt < 0) | (tgt >= c)
    assert np.all(got.transpose(0, 2, 3, 1)[bad] == 0)
    # nothing but ignored pixels: loss 0, gradient 0, no division by zero
    td2 = G.dev(np.full((n, h, w), 255, np.int32))
    capi.unet_softmax_xent_f32(G.p(ld), G.p(td2), G.p(gl), G.p(loss), n, c, h, w, None)
    assert G.host(loss)[0] == 0 and not G.host(gl).any()
    # labels that arrive as floats: NaN, infinities and values beyond int32 become "no class"
    f = np.array([0.0, 1.4, 3.6, np.nan, np.inf, -np.inf, 3e9, -3e9, 255.0], np.float32)
    out = torch.zeros(f.size, dtype=torch.int32, device=G.DEV)
    fd = G.dev(f)
    capi.unet_f32_to_i32(G.p(fd), G.p(out), f.size, None)
    assert out.cpu().tolist() == [0, 1, 4, -1, -1, -1, -1, -1, 255]


def test_adam(G, capi, oracle, golden):
    w = G.dev(golden["adam_w0"])
    st = torch.zeros(2 * w.numel(), device=G.DEV)
    for t, g in enumerate(golden["adam_grads"], start=1):
        gd = G.dev(g)
        capi.unet_update_weights_adam_f32(G.p(w), G.p(gd), G.p(st), 0.01, w.numel(), t, None)
        torch.cuda.synchronize()
    assert_close(G.host(w), golden["adam_w3"], what="adam weights")
    assert_close(G.host(st), golden["adam_state3"], what="adam state")
    rng = np.random.default_rng(11)
    n = 100003
    w0, g0 = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    st0 = np.abs(rng.standard_normal(2 * n)).astype(np.float32) * 0.01
    wd, sd, gd = G.dev(w0), G.dev(st0), G.dev(g0)
    capi.unet_update_weights_adam_f32(G.p(wd), G.p(gd), G.p(sd), 1e-3, n, 7, None)
    w_ref, st_ref = oracle.adam(w0, g0, st0, 1e-3, 7)
    assert_close(G.host(wd), w_ref, rtol=1e-5, what="adam vs oracle")
    assert_close(G.host(sd), st_ref, rtol=1e-5, what="adam state vs oracle")
    # vectors that do not start on a 16-byte boundary take the one-value-per-lane kernel for everything: the same
    # bits as the 16-byte kernel plus its tail
    f32 = lambda *parts: np.concatenate(parts).astype(np.float32)
    wu, su, gu = G.dev(f32([0], w0)), G.dev(f32([0, 0], st0)), G.dev(f32([0], g0))
    capi.unet_update_weights_adam_f32(G.p(wu) + 4, G.p(gu) + 4, G.p(su) + 8, 1e-3, n, 7, None)
    np.testing.assert_array_equal(G.host(wu)[1:], G.host(wd))
    np.testing.assert_array_equal(G.host(su)[2:], G.host(sd))


def test_init_and_nan_scan(G, capi):
    n = 1 << 18
    w = G.empty((n + 64,))
    w.zero_()
    capi.unet_init_he_f32(G.p(w), n, 576, 1234, None)
    h = G.host(w)
    assert np.all(h[n:] == 0)
    std = np.sqrt(2.0 / 576)
    assert abs(h[:n].mean()) < 4 * std / np.sqrt(n)
    assert abs(h[:n].std() / std - 1) < 0.02
    w2 = G.empty((n,))
    capi.unet_init_he_f32(G.p(w2), n, 576, 1234, None)
    assert_exact(G.host(w2), h[:n], what="same seed, same values")
    capi.unet_init_bn_f32(G.p(w2), 10, None)
    assert_exact(G.host(w2)[:10], np.array([1, 0] * 5, np.float32))
    flag = ctypes.c_int(-1)
    capi.unet_has_nans_f32(G.p(w), n, ctypes.byref(flag), None)
    assert flag.value == 0
    w[12345] = float("nan")
    capi.unet_has_nans_f32(G.p(w), n, ctypes.byref(flag), None)
    assert flag.value == 1


def test_runtime_memory_roundtrip(G, capi):
    ptr = ctypes.c_void_p()
    capi.unet_malloc(ctypes.byref(ptr), 4096)
    src = np.arange(1024, dtype=np.float32)
    back = np.full(1024, -1, np.float32)
    capi.unet_memcpy_d2h(back.ctypes.data, ptr, 4096, 1, None)
    assert np.all(back == 0), "unet_malloc must zero the allocation"
    capi.unet_memcpy_h2d(ptr, src.ctypes.data, 4096, None)
    capi.unet_memcpy_d2h(back.ctypes.data, ptr, 4096, 1, None)
    assert_exact(back, src)
    capi.unet_free(ptr)
    capi.unet_device_synchronize()
