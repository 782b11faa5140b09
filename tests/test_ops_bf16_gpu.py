"""GPU parity tests of the bf16 / NHWC fast path (MFMA implicit-GEMM
convolution, transposed convolution, weight gradients) against the CPU oracle.

Inputs are rounded to bf16 first, so both sides see identical operands; the
HIP side accumulates in fp32 and rounds outputs to bf16.  Bar: bf16 relative
<= 2e-2 (util.assert_close_bf16); fp32 outputs (weight gradients, statistics)
are held to 1e-3 of the tensor's magnitude.
"""
import numpy as np
import pytest
import torch

from util import assert_close, assert_close_bf16

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


@pytest.fixture(autouse=True)
def pipelined_kernel_for_small_shapes(capi):
    """The test shapes are small: lower the work threshold so that the pipelined 3x3 kernel runs wherever its
    geometry allows, as it does at training sizes (its output is bit-identical to the general kernel's)."""
    old = capi.lib().unet_conv3x3_pipeline_min_units(1)
    yield
    capi.lib().unet_conv3x3_pipeline_min_units(old)


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).bfloat16().float().numpy()


def pad32(c):
    return (c + 31) // 32 * 32


def to_nhwc(G, capi, x, ld=None, cpad=None):
    n, c, h, w = x.shape
    cpad = cpad or pad32(c)
    ld = ld or cpad
    xd = G.dev(x)
    out = torch.zeros((n, h, w, ld), dtype=torch.bfloat16, device=G.DEV)
    capi.unet_nchw_f32_to_nhwc_bf16(G.p(xd), G.p(out), n, c, h, w, ld, cpad, None)
    return out


def from_nhwc(G, capi, t, c):
    n, h, w, ld = t.shape
    out = G.empty((n, c, h, w))
    capi.unet_nhwc_bf16_to_nchw_f32(G.p(t), G.p(out), n, c, h, w, ld, None)
    return G.host(out)


def pack(G, capi, w_flat, ci, co, k):
    cip, cop = pad32(ci), pad32(co)
    wd = G.dev(w_flat)
    fwd = torch.zeros(k * k * cop * cip, dtype=torch.bfloat16, device=G.DEV)
    bwd = torch.zeros(k * k * cop * cip, dtype=torch.bfloat16, device=G.DEV)
    capi.unet_pack_weights_bf16(G.p(wd), G.p(fwd), G.p(bwd), ci, co, cip, cop, k, None)
    return fwd, bwd


def test_layout_round_trip(G, capi):
    rng = np.random.default_rng(0)
    x = bf16_round(rng.standard_normal((2, 5, 7, 9)))
    t = to_nhwc(G, capi, x, ld=40, cpad=32)
    th = t.float().cpu().numpy()
    np.testing.assert_array_equal(th[..., :5], x.transpose(0, 2, 3, 1))
    assert np.all(th[..., 5:32] == 0)
    np.testing.assert_array_equal(from_nhwc(G, capi, t, 5), x)


CONV_CASES = [
    # n, ci, co, h, w
    (2, 32, 64, 16, 16),
    (1, 64, 32, 20, 36),     # ragged tiles, BN=32 path
    (2, 96, 128, 8, 8),      # small image, 3 K chunks, 2 channel tiles
    (1, 3, 64, 33, 17),      # first-layer-like (C_in padded 3 -> 32)
    (2, 64, 1, 16, 24),      # last-layer-like (C_out padded 1 -> 32)
    (1, 128, 64, 48, 48),
    # enough tiles x channel blocks for the 16x16x32 main loop (forward and, for the last one, data gradient too)
    (1, 64, 128, 40, 72),    # ragged tiles, 2 K chunks, 2 channel blocks
    (1, 32, 64, 80, 80),     # one K chunk
    (1, 128, 64, 72, 72),    # 4 K chunks; data gradient with 2 channel blocks
    # the pipelined persistent kernel (images of 32 rows and more, 64-channel blocks): ragged 32x16 tiles, several
    # images per workgroup, weights resident in LDS (C_in = 64) and streamed per stage (C_in = 96: three chunks)
    (3, 64, 64, 50, 30),
    (2, 96, 192, 33, 47),
    (5, 64, 128, 64, 64),    # more tiles than workgroups per channel block
    # its 16-row tiles (images of 16 to 31 rows): resident weights and a ragged second tile row, five chunks and
    # two channel blocks on one tile per image, ragged in both directions
    (3, 64, 64, 20, 40),
    (2, 160, 128, 16, 16),
    (2, 96, 64, 31, 17),
]


def test_conv3x3_stat_rows_match_the_kernel_that_runs(G, capi):
    """The row count unet_conv3x3_stat_rows promises is the one the call writes, on either side of the pipelined
    kernel's work threshold; it stays within the documented bound; statistics together with accumulate (whose row
    count would differ) are rejected."""
    n, c, h, w = 4, 64, 32, 32
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).to(G.DEV).bfloat16()
    fwd, _ = pack(G, capi, bf16_round(rng.standard_normal(c * c * 9) * 0.05), c, c, 3)
    out = torch.zeros((n, h, w, c), dtype=torch.bfloat16, device=G.DEV)
    tiles = capi.lib().unet_conv3x3_stat_tiles(h, w)
    bound = max(n * tiles, 1024)
    sums = []
    old = capi.lib().unet_conv3x3_pipeline_min_units(-1)
    try:
        for thr in (1, 1 << 30):   # pipelined kernel / general kernel
            capi.lib().unet_conv3x3_pipeline_min_units(thr)
            rows = capi.lib().unet_conv3x3_stat_rows(n, h, w, c, c)
            assert 0 < rows <= bound
            guard = 8
            stats = torch.full((rows + guard, c, 2), 7.0, device=G.DEV)
            capi.unet_conv3x3_bf16(G.p(x), c, G.p(fwd), None, 0, G.p(out), c, G.p(stats), n, h, w, c, c, 0, None)
            st = stats.cpu().numpy()
            assert np.all(st[rows:] == 7.0), "rows beyond the promised count were written"
            sums.append(st[:rows].sum(axis=0))
            with pytest.raises(RuntimeError, match="accumulate"):
                capi.unet_conv3x3_bf16(G.p(x), c, G.p(fwd), None, 0, G.p(out), c, G.p(stats), n, h, w, c, c, 1, None)
    finally:
        capi.lib().unet_conv3x3_pipeline_min_units(old)
    assert_close(sums[0], sums[1], rtol=1e-5, atol_scale=1e-6, what="statistics of the two kernels")


@pytest.mark.parametrize("cfg", CONV_CASES)
def test_conv3x3_forward_dgrad_wgrad(G, capi, oracle, cfg):
    n, ci, co, h, w = cfg
    cip, cop = pad32(ci), pad32(co)
    rng = np.random.default_rng(ci * 7 + co)
    x = bf16_round(rng.standard_normal((n, ci, h, w)))
    wt = bf16_round(rng.standard_normal(ci * co * 9) * (1.0 / np.sqrt(9 * ci)))
    bias = (rng.standard_normal(co) * 0.5).astype(np.float32)
    w_and_b = np.concatenate([wt, bias]).astype(np.float32)
    dy = bf16_round(rng.standard_normal((n, co, h, w)))

    xt = to_nhwc(G, capi, x)
    fwd, bwd = pack(G, capi, wt, ci, co, 3)
    bd = G.dev(bias)
    rows = capi.lib().unet_conv3x3_stat_rows(n, h, w, cip, cop)
    out = torch.full((n, h, w, cop), float("nan"), dtype=torch.bfloat16, device=G.DEV)
    stats = G.empty((rows, cop, 2))
    capi.unet_conv3x3_bf16(G.p(xt), cip, G.p(fwd), G.p(bd), co, G.p(out), cop, G.p(stats), n, h, w, cip, cop, 0, None)
    y = from_nhwc(G, capi, out, co)
    y_ref = oracle.convolve(x, w_and_b, co, 3, 1, 1)
    assert_close_bf16(y, y_ref, what=f"conv3x3 bf16 forward {cfg}")
    outh = out.float().cpu().numpy()
    assert np.all(outh[..., co:] == 0), "padded output channels must be zero"
    # statistics are sums of the ROUNDED outputs
    st = G.host(stats).sum(axis=0)
    assert_close(st[:co, 0], y.sum(axis=(0, 2, 3)), rtol=1e-4, atol_scale=1e-5, what="conv stats sum")
    assert_close(st[:co, 1], (y.astype(np.float64) ** 2).sum(axis=(0, 2, 3)), rtol=1e-4, atol_scale=1e-5,
                 what="conv stats sum of squares")

    # data gradient = forward kernel over dy with the flipped / transposed weights
    dyt = to_nhwc(G, capi, dy)
    dx = torch.full((n, h, w, cip), float("nan"), dtype=torch.bfloat16, device=G.DEV)
    capi.unet_conv3x3_bf16(G.p(dyt), cop, G.p(bwd), None, 0, G.p(dx), cip, None, n, h, w, cop, cip, 0, None)
    dx_ref, dw_ref = oracle.convolve_backward(x, dy, w_and_b, 3, 1, 1)
    assert_close_bf16(from_nhwc(G, capi, dx, ci), dx_ref, what=f"conv3x3 bf16 dgrad {cfg}")
    # accumulate mode adds to what the buffer holds
    capi.unet_conv3x3_bf16(G.p(dyt), cop, G.p(bwd), None, 0, G.p(dx), cip, None, n, h, w, cop, cip, 1, None)
    assert_close_bf16(from_nhwc(G, capi, dx, ci), 2 * dx_ref, what=f"conv3x3 bf16 dgrad accumulate {cfg}")

    # weight gradient, fp32 in the reference layout
    gw = G.empty((ci * co * 9,))
    capi.unet_conv3x3_wgrad_bf16(G.p(dyt), cop, G.p(xt), cip, G.p(gw), n, h, w, cip, cop, ci, co, None)
    ref = dw_ref[:ci * co * 9]
    got = G.host(gw)
    assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max(), (np.abs(got - ref).max(), np.abs(ref).max())


@pytest.mark.parametrize("cfg", [(3, 256, 256, 50, 70), (1, 512, 512, 64, 128), (2, 64, 192, 72, 80)])
def test_conv3x3_wgrad_long_pixel_slices(G, capi, cfg):
    """Weight gradient where one workgroup walks many 8 x 16 pixel chunks in a row (all three stage buffers of
    its load pipeline get reused several times), with ragged image edges, across image boundaries, and with a
    slice that ends before its last trip: against torch's fp32 convolution gradient on the CPU."""
    n, ci, co, h, w = cfg
    rng = np.random.default_rng(ci + co + h)
    x = bf16_round(rng.standard_normal((n, ci, h, w)))
    dy = bf16_round(rng.standard_normal((n, co, h, w)))
    xt, dyt = to_nhwc(G, capi, x), to_nhwc(G, capi, dy)
    gw = G.empty((ci * co * 9,))
    capi.unet_conv3x3_wgrad_bf16(G.p(dyt), co, G.p(xt), ci, G.p(gw), n, h, w, ci, co, ci, co, None)
    got = G.host(gw).reshape(ci, co, 3, 3)
    ref = torch.nn.grad.conv2d_weight(torch.from_numpy(x), (co, ci, 3, 3), torch.from_numpy(dy), padding=1)
    ref = ref.permute(1, 0, 2, 3).numpy()
    assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max(), (np.abs(got - ref).max(), np.abs(ref).max())
    again = G.empty((ci * co * 9,))
    capi.unet_conv3x3_wgrad_bf16(G.p(dyt), co, G.p(xt), ci, G.p(again), n, h, w, ci, co, ci, co, None)
    np.testing.assert_array_equal(G.host(again).reshape(ci, co, 3, 3), got)


@pytest.mark.parametrize("case", [(8, 64, 64, 256), (16, 512, 1024, 16), (4, 64, 64, 256), (4, 256, 256, 64),
                                  (6, 128, 128, 72), (8, 64, 192, 40)])
def test_pipelined_kernel_is_bit_identical_to_the_general_kernel_at_training_sizes(G, capi, case):
    """The two 3x3 kernels on the same operands, forward (with statistics) and data gradient: every output
    bit equal.  At 256 x 256 with batch 8 the data gradient is large enough (67 MB) for the pipelined kernel's
    non-temporal stores, and every workgroup walks several tiles in reversed order; 16 x 16 with 1024 channels
    is the 16-row tile variant at the size the benchmark graph uses it; batch 4 at 256 x 256 takes 16-row tiles
    because 32-row tiles would leave a workgroup two of them (resident weights); batch 4 at 64 x 64 with 256
    channels because there are fewer 32-row tiles than workgroups (streamed weights); 72 x 72 and 40 x 40 have
    ragged tiles on both sides and tile counts per image (15, 6) that the workgroups' strides do not divide, so
    the tile walk carries over rows and images unevenly and the masked statistics path runs."""
    n, ci, co, h = case
    w = h
    gen = torch.Generator(device="cpu").manual_seed(h)
    x = torch.randn((n, h, w, ci), generator=gen).to(G.DEV).bfloat16()
    dy = torch.randn((n, h, w, co), generator=gen).to(G.DEV).bfloat16()
    wt = (torch.randn(ci * co * 9, generator=gen) / np.sqrt(9 * ci)).numpy().astype(np.float32)
    fwd, bwd = pack(G, capi, wt, ci, co, 3)
    bias = G.dev((np.arange(co) % 7 - 3).astype(np.float32) * 0.1)
    lib = capi.lib()

    def run():
        rows = lib.unet_conv3x3_stat_rows(n, h, w, ci, co)
        out = torch.full((n, h, w, co), float("nan"), dtype=torch.bfloat16, device=G.DEV)
        stats = G.empty((rows, co, 2))
        capi.unet_conv3x3_bf16(G.p(x), ci, G.p(fwd), G.p(bias), co, G.p(out), co, G.p(stats), n, h, w, ci, co, 0, None)
        dx = torch.full((n, h, w, ci), float("nan"), dtype=torch.bfloat16, device=G.DEV)
        capi.unet_conv3x3_bf16(G.p(dy), co, G.p(bwd), None, 0, G.p(dx), ci, None, n, h, w, co, ci, 0, None)
        torch.cuda.synchronize()
        return out, dx, stats.double().sum(dim=0)

    old = lib.unet_conv3x3_pipeline_min_units(128)        # the threshold of training runs: the pipelined kernel
    try:
        out_p, dx_p, st_p = run()
        lib.unet_conv3x3_pipeline_min_units(1 << 30)      # nothing is large enough: the general kernel
        out_g, dx_g, st_g = run()
    finally:
        lib.unet_conv3x3_pipeline_min_units(old)
    assert torch.equal(out_p.view(torch.int16), out_g.view(torch.int16))
    assert torch.equal(dx_p.view(torch.int16), dx_g.view(torch.int16))
    assert torch.allclose(st_p, st_g, rtol=1e-5, atol=1e-3 * float(st_g.abs().max()))


def test_conv3x3_reads_channel_slices(G, capi, oracle):
    """Input and output tensors that are channel slices of wider buffers
    (pixel stride > channel count): the zero-copy concat layout."""
    n, ci, co, h, w = 1, 32, 32, 16, 16
    rng = np.random.default_rng(3)
    x = bf16_round(rng.standard_normal((n, ci, h, w)))
    wt = bf16_round(rng.standard_normal(ci * co * 9) * 0.1)
    big_in = torch.full((n, h, w, 96), 7.0, dtype=torch.bfloat16, device=G.DEV)
    big_in[..., 64:96] = torch.from_numpy(x.transpose(0, 2, 3, 1)).to(G.DEV).bfloat16()
    big_out = torch.full((n, h, w, 64), 5.0, dtype=torch.bfloat16, device=G.DEV)
    fwd, _ = pack(G, capi, wt, ci, co, 3)
    capi.unet_conv3x3_bf16(big_in.data_ptr() + 64 * 2, 96, G.p(fwd), None, 0, big_out.data_ptr() + 32 * 2, 64, None,
                           n, h, w, 32, 32, 0, None)
    res = big_out.float().cpu().numpy()
    assert np.all(res[..., :32] == 5.0), "channels outside the slice must be untouched"
    y_ref = oracle.convolve(x, np.concatenate([wt, np.zeros(co, np.float32)]), co, 3, 1, 1)
    assert_close_bf16(res[..., 32:].transpose(0, 3, 1, 2), y_ref, what="conv into a channel slice")


# From the third on: shapes the matrix-product kernel serves (input channels in multiples of 64; the data gradient
# when its output channels come in multiples of 128): ragged images, one to six reduction chunks (odd counts too),
# column tiles that span four, two and one output phase, padded output channels.
@pytest.mark.parametrize("cfg", [(2, 64, 32, 8, 8), (1, 32, 64, 20, 12), (1, 128, 64, 16, 16), (2, 64, 64, 20, 12),
                                 (1, 128, 64, 20, 28), (3, 192, 128, 16, 32), (2, 256, 128, 24, 16), (1, 128, 40, 16, 16),
                                 (2, 384, 192, 8, 40)])
def test_tconv2x2(G, capi, oracle, cfg):
    n, ci, co, h, w = cfg
    cip, cop = pad32(ci), pad32(co)
    rng = np.random.default_rng(ci + co)
    x = bf16_round(rng.standard_normal((n, ci, h, w)))
    wt = bf16_round(rng.standard_normal(ci * co * 4) * (1.0 / np.sqrt(ci)))
    bias = (rng.standard_normal(co) * 0.5).astype(np.float32)
    w_and_b = np.concatenate([wt, bias]).astype(np.float32)
    dy = bf16_round(rng.standard_normal((n, co, 2 * h, 2 * w)))
    xt, dyt = to_nhwc(G, capi, x), to_nhwc(G, capi, dy)
    fwd, bwd = pack(G, capi, wt, ci, co, 2)
    bd = G.dev(bias)
    tiles = capi.lib().unet_conv3x3_stat_tiles(h, w)
    out = torch.full((n, 2 * h, 2 * w, cop), float("nan"), dtype=torch.bfloat16, device=G.DEV)
    stats = G.empty((4, n, tiles, cop, 2))
    capi.unet_tconv2x2_bf16(G.p(xt), cip, G.p(fwd), G.p(bd), co, G.p(out), cop, G.p(stats), n, h, w, cip, cop, None)
    y = from_nhwc(G, capi, out, co)
    assert_close_bf16(y, oracle.upsample(x, w_and_b, co, 2), what=f"tconv forward {cfg}")
    st = G.host(stats).sum(axis=(0, 1, 2))
    assert_close(st[:co, 0], y.sum(axis=(0, 2, 3)), rtol=1e-4, atol_scale=1e-5, what="tconv stats sum")
    assert_close(st[:co, 1], (y.astype(np.float64) ** 2).sum(axis=(0, 2, 3)), rtol=1e-4, atol_scale=1e-5, what="tconv stats sum of squares")
    dx_ref, dw_ref = oracle.upsample_backward(x, w_and_b, dy, 2)
    dx = torch.full((n, h, w, cip), float("nan"), dtype=torch.bfloat16, device=G.DEV)
    capi.unet_tconv2x2_dgrad_bf16(G.p(dyt), cop, G.p(bwd), G.p(dx), cip, n, h, w, cip, cop, 0, None)
    assert_close_bf16(from_nhwc(G, capi, dx, ci), dx_ref, what=f"tconv dgrad {cfg}")
    gw = G.empty((ci * co * 4,))
    capi.unet_tconv2x2_wgrad_bf16(G.p(dyt), cop, G.p(xt), cip, G.p(gw), n, h, w, cip, cop, ci, co, None)
    ref = dw_ref[:ci * co * 4]
    got = G.host(gw)
    assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max(), (np.abs(got - ref).max(), np.abs(ref).max())


@pytest.mark.parametrize("cfg", [(2, 64, 32, 8, 8), (1, 128, 64, 20, 28), (2, 256, 128, 24, 16)])
def test_tconv2x2_dgrad_accumulates_into_a_gradient(G, capi, cfg):
    """accumulate = 1 (a second consumer of the same tensor): the stored sum is bf16(bf16(dx) + old), whichever of
    the two data-gradient kernels the shape selects (general for the first shape, matrix product for the others)."""
    n, ci, co, h, w = cfg
    rng = np.random.default_rng(17)
    wt = bf16_round(rng.standard_normal(ci * co * 4) * (1.0 / np.sqrt(co)))
    _, bwd = pack(G, capi, wt, ci, co, 2)
    dyt = torch.from_numpy(rng.standard_normal((n, 2 * h, 2 * w, co)).astype(np.float32)).to(G.DEV).bfloat16()
    old = torch.from_numpy(rng.standard_normal((n, h, w, ci)).astype(np.float32)).to(G.DEV).bfloat16()
    dx = torch.full((n, h, w, ci), float("nan"), dtype=torch.bfloat16, device=G.DEV)
    capi.unet_tconv2x2_dgrad_bf16(G.p(dyt), co, G.p(bwd), G.p(dx), ci, n, h, w, ci, co, 0, None)
    acc = old.clone()
    capi.unet_tconv2x2_dgrad_bf16(G.p(dyt), co, G.p(bwd), G.p(acc), ci, n, h, w, ci, co, 1, None)
    torch.cuda.synchronize()
    assert torch.isfinite(dx.float()).all() and dx.float().abs().max() > 0
    assert torch.equal(acc, (dx.float() + old.float()).bfloat16())


# ------------------------------------------------ fused BN / activation ----
def nhwc_tensor(G, x, ld=None):
    """x [N,C,H,W] float (bf16-representable) -> NHWC bf16 device tensor with pixel stride ld."""
    n, c, h, w = x.shape
    ld = ld or c
    t = torch.zeros((n, h, w, ld), dtype=torch.bfloat16, device=G.DEV)
    t[..., :c] = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).to(G.DEV).bfloat16()
    return t


def nchw(t, c):
    return t[..., :c].float().cpu().numpy().transpose(0, 3, 1, 2)


# the last case is large enough (>= 32 MB per pass) for the streaming variants of the kernels: non-temporal loads
# and stores, one contiguous piece of the tensor per workgroup
@pytest.mark.parametrize("cfg", [(2, 64, 16, 16, 0), (1, 32, 9, 7, 0), (3, 96, 8, 8, 0), (2, 32, 8, 8, 1),
                                 (1, 256, 4, 4, 2), (2, 24, 6, 6, 0), (3, 64, 250, 254, 0),
                                 # LEAKY_ReLU (slope 0.1) and TANH, through the entry points that take the slope
                                 (2, 64, 16, 16, 3), (1, 32, 9, 7, 4), (2, 40, 8, 8, 3), (3, 64, 250, 254, 4)])
def test_bn_act_forward_backward(G, capi, oracle, cfg):
    n, c, h, w, act = cfg
    slope = 0.1
    fwd_fn = [oracle.relu, oracle.sigmoid, lambda v: v, lambda v: oracle.leaky_relu(v, slope), oracle.tanh][act]

    def bwd_fn(dy, bn):
        if act == 0:
            return oracle.relu_backward(dy, bn)
        if act == 1:
            return oracle.sigmoid_backward(dy, oracle.sigmoid(bn))
        if act == 3:
            return oracle.leaky_relu_backward(dy, bn, slope)
        if act == 4:
            return oracle.tanh_backward(dy, oracle.tanh(bn))
        return dy

    def bn_act(*a):
        return capi.unet_bn_act_bf16(*a[:-1], act, None) if act <= 2 else capi.unet_bn_act_ex_bf16(*a[:-1], act, slope, None)

    def bn_act_backward(*a):
        if act <= 2:
            return capi.unet_bn_act_backward_bf16(*a, act, None)
        return capi.unet_bn_act_backward_ex_bf16(*a, act, slope, None)

    cop = (c + 7) // 8 * 8
    rng = np.random.default_rng(c * 10 + act)
    z = bf16_round(rng.standard_normal((n, c, h, w)) * 1.3 + 0.4)
    gb = (rng.standard_normal(2 * c) * 0.5 + np.tile([1.0, 0.0], c)).astype(np.float32)
    run_in = np.abs(rng.standard_normal(2 * c)).astype(np.float32)
    dy1 = bf16_round(rng.standard_normal((n, c, h, w)))
    dy2 = bf16_round(rng.standard_normal((n, c, h, w)))
    M = n * h * w
    # partial sums as the convolution epilogue would deliver them (two slices to exercise the reduction)
    half = z[:, :, : h // 2], z[:, :, h // 2:]
    partial = np.zeros((2, cop, 2), np.float32)
    for i, part in enumerate(half):
        partial[i, :c, 0] = part.sum(axis=(0, 2, 3))
        partial[i, :c, 1] = (part.astype(np.float64) ** 2).sum(axis=(0, 2, 3))
    zt = nhwc_tensor(G, z, cop)
    pd, gbd, rund = G.dev(partial), G.dev(gb), G.dev(run_in)
    stats, coef = G.empty((2 * c,)), G.empty((capi.lib().unet_bn_coef_floats(cop),))
    capi.unet_bn_finalize_bf16(G.p(pd), 2, c, cop, M, G.p(gbd), G.p(stats), G.p(rund), G.p(coef), None)
    y = torch.full((n, h, w, cop + 8), 3.0, dtype=torch.bfloat16, device=G.DEV)   # y is a slice of a wider buffer
    bn_act(G.p(zt), cop, y.data_ptr() + 16, cop + 8, G.p(coef), cop, n * h * w, None)
    bn_ref, stats_ref, run_ref = oracle.bn_train(z, gb, run_in)
    y_ref = fwd_fn(bn_ref)
    yh = y.float().cpu().numpy()
    assert np.all(yh[..., :8] == 3.0)
    assert_close_bf16(yh[..., 8:8 + c].transpose(0, 3, 1, 2), y_ref, what=f"bn+act forward {cfg}")
    assert_close(G.host(stats), stats_ref, rtol=1e-5, what="batch stats")
    assert_close(G.host(rund), run_ref, rtol=1e-5, atol_scale=1e-6, what="running stats")

    # backward: fan-in of two gradients, activation backward, BN backward, bias gradient
    d1, d2 = nhwc_tensor(G, dy1, cop), nhwc_tensor(G, dy2, cop)
    dz = torch.full((n, h, w, cop), float("nan"), dtype=torch.bfloat16, device=G.DEV)
    ggb, gbias = G.empty((2 * c,)), G.empty((c,))
    bn_act_backward(G.p(d1), cop, G.p(d2), cop, G.p(zt), cop, G.p(coef), G.p(gbd), G.p(dz), cop,
                    G.p(ggb), G.p(gbias), c, cop, n * h * w)
    dy = dy1 + dy2
    g = bwd_fn(dy, bn_ref)
    dz_ref, ggb_ref = oracle.bn_backward(z, g, gb, stats_ref)
    got = nchw(dz, c)
    assert_close_bf16(got, dz_ref, what=f"bn+act backward {cfg}")
    gg = G.host(ggb)
    assert np.abs(gg - ggb_ref).max() <= 2e-3 * np.abs(ggb_ref).max() + 1e-4, (gg, ggb_ref)
    assert_close(G.host(gbias), got.astype(np.float64).sum(axis=(0, 2, 3)), rtol=1e-4, atol_scale=1e-4,
                 what="bias gradient = sum of dz")
    # single-source variant
    bn_act_backward(G.p(d1), cop, None, 0, G.p(zt), cop, G.p(coef), G.p(gbd), G.p(dz), cop,
                    G.p(ggb), None, c, cop, n * h * w)
    g1 = bwd_fn(dy1, bn_ref)
    dz1_ref, _ = oracle.bn_backward(z, g1, gb, stats_ref)
    assert_close_bf16(nchw(dz, c), dz1_ref, what=f"bn+act backward, one source {cfg}")


def test_bn_inference_coef(G, capi, oracle):
    n, c, h, w = 2, 32, 5, 6
    rng = np.random.default_rng(2)
    z = bf16_round(rng.standard_normal((n, c, h, w)))
    gb = rng.standard_normal(2 * c).astype(np.float32)
    run = np.abs(rng.standard_normal(2 * c)).astype(np.float32) + 0.1
    zt, gbd, rund = nhwc_tensor(G, z), G.dev(gb), G.dev(run)
    coef = G.empty((capi.lib().unet_bn_coef_floats(c),))
    capi.unet_bn_inference_coef_bf16(G.p(gbd), G.p(rund), c, c, G.p(coef), None)
    y = torch.empty((n, h, w, c), dtype=torch.bfloat16, device=G.DEV)
    capi.unet_bn_act_bf16(G.p(zt), c, G.p(y), c, G.p(coef), c, n * h * w, 2, None)
    assert_close_bf16(nchw(y, c), oracle.bn_inference(z, gb, run), what="bn inference bf16")


@pytest.mark.parametrize("cfg", [(2, 64, 16, 16), (1, 32, 6, 10), (3, 8, 2, 2), (1, 128, 32, 32)])
def test_max_pool_bf16(G, capi, oracle, cfg):
    n, c, h, w = cfg
    rng = np.random.default_rng(c + h)
    x = bf16_round(np.maximum(np.round(rng.standard_normal((n, c, h, w)) * 4) / 4, 0))   # post-ReLU-like, many ties
    dy = bf16_round(rng.standard_normal((n, c, h // 2, w // 2)))
    xt, dyt = nhwc_tensor(G, x, c + 8), nhwc_tensor(G, dy)
    y = torch.empty((n, h // 2, w // 2, c), dtype=torch.bfloat16, device=G.DEV)
    mask = torch.full((n, h // 2, w // 2, c), 255, dtype=torch.uint8, device=G.DEV)
    capi.unet_max_pool2x2_bf16(G.p(xt), c + 8, G.p(y), c, G.p(mask), n, h, w, c, None)
    y_ref, tm_ref = oracle.maxpool(x, 2, 2, want_tiemask=True)
    np.testing.assert_array_equal(nchw(y, c), y_ref)
    np.testing.assert_array_equal(mask.cpu().numpy().transpose(0, 3, 1, 2), tm_ref)
    dx = torch.full((n, h, w, c), float("nan"), dtype=torch.bfloat16, device=G.DEV)
    capi.unet_max_pool2x2_backward_bf16(G.p(dyt), c, G.p(mask), G.p(dx), c, n, h, w, c, 0, None)
    dx_ref = oracle.maxpool_backward(dy, x, y_ref, 2)
    np.testing.assert_array_equal(nchw(dx, c), dx_ref)
    capi.unet_max_pool2x2_backward_bf16(G.p(dyt), c, G.p(mask), G.p(dx), c, n, h, w, c, 1, None)
    np.testing.assert_array_equal(nchw(dx, c), bf16_round(2 * dx_ref))


# the last case is large enough (>= 32 MB per pass) for the non-temporal variants of the kernels
@pytest.mark.parametrize("cfg", [(2, 64, 16, 16, 0, True), (1, 32, 6, 10, 0, False), (3, 96, 8, 12, 1, True),
                                 (2, 128, 40, 24, 0, True), (1, 8, 2, 2, 2, False), (3, 64, 250, 254, 0, True)])
def test_bn_act_with_fused_max_pool_forward_and_backward(G, capi, oracle, cfg):
    """The skip layers' fused passes (activation + pool; pool backward gathered inside the BN backward): the
    forward must give bit for bit what the separate calls give, the backward the same values up to the order of
    the fp32 sums, and both the oracle's pool / BN results on the same operands."""
    n, c, h, w, act, with_dy = cfg
    rng = np.random.default_rng(c + 3 * h + act)
    z = bf16_round(np.round(rng.standard_normal((n, c, h, w)) * 3) / 3 + 0.2)   # coarse values: ties in the windows
    gb = (rng.standard_normal(2 * c) * 0.5 + np.tile([1.0, 0.0], c)).astype(np.float32)
    M = n * h * w
    partial = np.zeros((1, c, 2), np.float32)
    partial[0, :, 0] = z.sum(axis=(0, 2, 3))
    partial[0, :, 1] = (z.astype(np.float64) ** 2).sum(axis=(0, 2, 3))
    zt, pd, gbd, rund = nhwc_tensor(G, z, c), G.dev(partial), G.dev(gb), G.dev(np.ones(2 * c, np.float32))
    stats, coef = G.empty((2 * c,)), G.empty((capi.lib().unet_bn_coef_floats(c),))
    capi.unet_bn_finalize_bf16(G.p(pd), 1, c, c, M, G.p(gbd), G.p(stats), G.p(rund), G.p(coef), None)
    ld = c + 8                                                                     # y lives in a wider buffer
    y_sep = torch.zeros((n, h, w, ld), dtype=torch.bfloat16, device=G.DEV)
    y_fus = torch.zeros_like(y_sep)
    p_sep = torch.empty((n, h // 2, w // 2, c), dtype=torch.bfloat16, device=G.DEV)
    p_fus = torch.full_like(p_sep, float("nan"))
    m_sep = torch.empty((n, h // 2, w // 2, c), dtype=torch.uint8, device=G.DEV)
    m_fus = torch.full_like(m_sep, 255)
    capi.unet_bn_act_bf16(G.p(zt), c, G.p(y_sep), ld, G.p(coef), c, M, act, None)
    capi.unet_max_pool2x2_bf16(G.p(y_sep), ld, G.p(p_sep), c, G.p(m_sep), n, h, w, c, None)
    capi.unet_bn_act_pool_bf16(G.p(zt), c, G.p(y_fus), ld, G.p(coef), G.p(p_fus), c, G.p(m_fus), n, h, w, c, act, None)
    assert torch.equal(y_sep, y_fus) and torch.equal(p_sep, p_fus) and torch.equal(m_sep, m_fus)
    y_host = y_fus.float().cpu().numpy()[..., :c].transpose(0, 3, 1, 2)
    p_ref, tm_ref = oracle.maxpool(y_host, 2, 2, want_tiemask=True)
    np.testing.assert_array_equal(nchw(p_fus, c), p_ref)
    np.testing.assert_array_equal(m_fus.cpu().numpy().transpose(0, 3, 1, 2), tm_ref)

    # backward: (optional) full-resolution gradient + the pooled gradient routed through the tie mask
    dy = bf16_round(rng.standard_normal((n, c, h, w)))
    dp = bf16_round(rng.standard_normal((n, c, h // 2, w // 2)))
    dyt, dpt = nhwc_tensor(G, dy, c), nhwc_tensor(G, dp, c)
    scattered = torch.empty((n, h, w, c), dtype=torch.bfloat16, device=G.DEV)
    capi.unet_max_pool2x2_backward_bf16(G.p(dpt), c, G.p(m_sep), G.p(scattered), c, n, h, w, c, 0, None)
    dz_sep = torch.empty((n, h, w, c), dtype=torch.bfloat16, device=G.DEV)
    dz_fus = torch.full_like(dz_sep, float("nan"))
    ggb_sep, ggb_fus, gbias_sep, gbias_fus = G.empty((2 * c,)), G.empty((2 * c,)), G.empty((c,)), G.empty((c,))
    if with_dy:
        capi.unet_bn_act_backward_bf16(G.p(dyt), c, G.p(scattered), c, G.p(zt), c, G.p(coef), G.p(gbd), G.p(dz_sep), c,
                                       G.p(ggb_sep), G.p(gbias_sep), c, c, M, act, None)
    else:
        capi.unet_bn_act_backward_bf16(G.p(scattered), c, None, 0, G.p(zt), c, G.p(coef), G.p(gbd), G.p(dz_sep), c,
                                       G.p(ggb_sep), G.p(gbias_sep), c, c, M, act, None)
    capi.unet_bn_act_backward_pooled_bf16(G.p(dyt) if with_dy else None, c, G.p(dpt), c, G.p(m_fus), n, h, w, G.p(zt), c,
                                          G.p(coef), G.p(gbd), G.p(dz_fus), c, G.p(ggb_fus), G.p(gbias_fus), c, c, act,
                                          None)
    # the fused passes add per pooling window, the separate ones per pixel: same sums, another order
    assert_close(G.host(ggb_fus), G.host(ggb_sep), rtol=1e-5, atol_scale=1e-5, what="dgamma/dbeta fused vs separate")
    assert_close_bf16(dz_fus.float().cpu().numpy(), dz_sep.float().cpu().numpy(), what="dz fused vs separate")
    assert_close(G.host(gbias_fus), nchw(dz_fus, c).astype(np.float64).sum(axis=(0, 2, 3)), rtol=1e-4, atol_scale=1e-4,
                 what="bias gradient = sum of dz")
    # and against the oracle
    bn_ref, stats_ref, _ = oracle.bn_train(z, gb, np.ones(2 * c, np.float32))
    g_in = oracle.maxpool_backward(dp, y_host, p_ref, 2) + (dy if with_dy else 0)
    if act == 0:
        g = oracle.relu_backward(g_in, bn_ref)
    elif act == 1:
        g = oracle.sigmoid_backward(g_in, oracle.sigmoid(bn_ref))
    else:
        g = g_in
    dz_ref, ggb_ref = oracle.bn_backward(z, g.astype(np.float32), gb, stats_ref)
    assert_close_bf16(nchw(dz_fus, c), dz_ref, what=f"pooled bn+act backward {cfg}")
    gg = G.host(ggb_fus)
    assert np.abs(gg - ggb_ref).max() <= 2e-3 * np.abs(ggb_ref).max() + 1e-4


@pytest.mark.parametrize("cfg", [("conv", 2, 64, 64, 24, 40), ("conv", 1, 32, 96, 16, 16), ("tconv", 2, 64, 32, 8, 12),
                                 ("tconv", 2, 128, 64, 20, 28), ("tconv", 1, 256, 128, 16, 16),  # the matrix-product kernel
                                 ("conv", 1, 64, 64, 72, 80),    # the pipelined kernel, weights resident
                                 ("conv", 3, 128, 192, 40, 48),   # the pipelined kernel, six K chunks, two channel blocks
                                 ("conv", 2, 128, 128, 16, 32)])  # its 16-row tiles
def test_dgrad_with_fused_bn_backward_reduction(G, capi, cfg):
    """The data-gradient kernels that also reduce the parent's batch-norm backward over the pixels:
    dx is bit-identical to the plain data gradient, and dz / dgamma / dbeta / dbias agree with the
    separate reduce + apply path (same sums, different order of addition)."""
    kind, n, cp, cn, h, w = cfg          # cp: parent's channels (= dx channels), cn: this layer's channels
    rng = np.random.default_rng(11)
    k = 3 if kind == "conv" else 2
    hz, wz = (h, w) if kind == "conv" else (2 * h, 2 * w)
    dzn = torch.from_numpy(rng.standard_normal((n, hz, wz, cn)).astype(np.float32)).to(G.DEV).bfloat16()
    wflat = (rng.standard_normal(cp * cn * k * k + cn) * 0.1).astype(np.float32)
    _, wb = pack(G, capi, wflat, cp, cn, k)
    zp = torch.from_numpy(rng.standard_normal((n, h, w, cp)).astype(np.float32)).to(G.DEV).bfloat16()
    # parent's BN records {a, b, mean, invb} and gamma/beta
    gb = G.dev((rng.random(2 * cp) + 0.5).astype(np.float32))
    coef_h = np.stack([rng.random(cp) + 0.5, rng.standard_normal(cp) * 0.3, rng.standard_normal(cp) * 0.1,
                       rng.random(cp) + 0.7], axis=1).astype(np.float32)
    coef = G.dev(coef_h.reshape(-1))
    pixels = n * h * w
    tiles = capi.lib().unet_conv3x3_stat_tiles(h, w)
    # rows of partial sums the fused data-gradient entry point writes
    nrows = capi.lib().unet_conv3x3_stat_rows(n, h, w, cn, cp) if kind == "conv" else n * tiles

    def run(fused):
        dx = torch.zeros((n, h, w, cp), dtype=torch.bfloat16, device=G.DEV)
        dzp = torch.zeros((n, h, w, cp), dtype=torch.bfloat16, device=G.DEV)
        ggb, gbias = G.empty((2 * cp,)), G.empty((cp,))
        if fused:
            parts = G.empty((nrows * cp * 2,))
            if kind == "conv":
                capi.unet_conv3x3_dgrad_bnred_bf16(G.p(dzn), cn, G.p(wb), G.p(dx), cp, G.p(zp), cp, G.p(coef), G.p(parts), 0,
                                                   n, h, w, cn, cp, None)
            else:
                capi.unet_tconv2x2_dgrad_bnred_bf16(G.p(dzn), cn, G.p(wb), G.p(dx), cp, G.p(zp), cp, G.p(coef),
                                                    G.p(parts), n, h, w, cp, cn, None)
            capi.unet_bn_act_backward_from_partials_bf16(G.p(dx), cp, G.p(zp), cp, G.p(coef), G.p(gb), G.p(parts),
                                                         nrows, G.p(dzp), cp, G.p(ggb), G.p(gbias), cp, cp, pixels, None)
        else:
            if kind == "conv":
                capi.unet_conv3x3_bf16(G.p(dzn), cn, G.p(wb), None, 0, G.p(dx), cp, None, n, h, w, cn, cp, 0, None)
            else:
                capi.unet_tconv2x2_dgrad_bf16(G.p(dzn), cn, G.p(wb), G.p(dx), cp, n, h, w, cp, cn, 0, None)
            capi.unet_bn_act_backward_bf16(G.p(dx), cp, None, 0, G.p(zp), cp, G.p(coef), G.p(gb), G.p(dzp), cp, G.p(ggb),
                                           G.p(gbias), cp, cp, pixels, 0, None)
        torch.cuda.synchronize()
        return dx.float().cpu().numpy(), dzp.float().cpu().numpy(), G.host(ggb), G.host(gbias)

    dx_a, dz_a, ggb_a, gbias_a = run(False)
    dx_b, dz_b, ggb_b, gbias_b = run(True)
    assert np.abs(dx_a).max() > 0
    np.testing.assert_array_equal(dx_a, dx_b)
    assert_close(ggb_b, ggb_a, rtol=1e-4, atol_scale=1e-5, what="dgamma/dbeta")
    # dz is bf16: the two paths may differ by one rounding where the fp32 value sits on a boundary, and the bias
    # gradient is the sum of the rounded values, so it may differ by a few units in the last place of the
    # largest dz
    assert_close_bf16(dz_b, dz_a, what="dz")
    ulp = float(np.abs(dz_a).max()) * 2.0 ** -8
    assert np.abs(gbias_b - gbias_a).max() <= 1e-3 * np.abs(gbias_a).max() + 4 * ulp, np.abs(gbias_b - gbias_a).max()


@pytest.mark.parametrize("hw", [(24, 16), (40, 48)])   # the general kernel; the pipelined one
def test_dgrad_fused_reduction_on_second_half_of_a_concat(G, capi, hw):
    """bn_first_channel = 64 of 128 output channels: the first half of dx is a plain data gradient,
    the second half also yields the BN-backward partial sums of the layer that produced it."""
    n, cn, (h, w), c1, c2 = 2, 64, hw, 64, 64
    cp = c1 + c2
    rng = np.random.default_rng(12)
    dzn = torch.from_numpy(rng.standard_normal((n, h, w, cn)).astype(np.float32)).to(G.DEV).bfloat16()
    wflat = (rng.standard_normal(cp * cn * 9 + cn) * 0.1).astype(np.float32)
    _, wb = pack(G, capi, wflat, cp, cn, 3)
    zu = torch.from_numpy(rng.standard_normal((n, h, w, c2)).astype(np.float32)).to(G.DEV).bfloat16()
    gb = G.dev((rng.random(2 * c2) + 0.5).astype(np.float32))
    coef = G.dev(np.stack([rng.random(c2) + 0.5, rng.standard_normal(c2) * 0.3, rng.standard_normal(c2) * 0.1,
                           rng.random(c2) + 0.7], axis=1).astype(np.float32).reshape(-1))
    pixels, nrows = n * h * w, capi.lib().unet_conv3x3_stat_rows(n, h, w, cn, cp)
    dx_a = torch.zeros((n, h, w, cp), dtype=torch.bfloat16, device=G.DEV)
    dx_b = torch.zeros_like(dx_a)
    capi.unet_conv3x3_bf16(G.p(dzn), cn, G.p(wb), None, 0, G.p(dx_a), cp, None, n, h, w, cn, cp, 0, None)
    parts = G.empty((nrows * c2 * 2,))
    capi.unet_conv3x3_dgrad_bnred_bf16(G.p(dzn), cn, G.p(wb), G.p(dx_b), cp, G.p(zu), c2, G.p(coef), G.p(parts), c1,
                                       n, h, w, cn, cp, None)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dx_a.float().cpu().numpy(), dx_b.float().cpu().numpy())
    out = {}
    for name, fused in (("sep", False), ("fused", True)):
        dzu = torch.zeros((n, h, w, c2), dtype=torch.bfloat16, device=G.DEV)
        ggb, gbias = G.empty((2 * c2,)), G.empty((c2,))
        dy_slice = dx_a.data_ptr() + 2 * c1         # channels [c1, c1+c2) of the concat gradient, pixel stride cp
        if fused:
            capi.unet_bn_act_backward_from_partials_bf16(dy_slice, cp, G.p(zu), c2, G.p(coef), G.p(gb), G.p(parts),
                                                         nrows, G.p(dzu), c2, G.p(ggb), G.p(gbias), c2, c2, pixels, None)
        else:
            capi.unet_bn_act_backward_bf16(dy_slice, cp, None, 0, G.p(zu), c2, G.p(coef), G.p(gb), G.p(dzu), c2, G.p(ggb),
                                           G.p(gbias), c2, c2, pixels, 0, None)
        torch.cuda.synchronize()
        out[name] = (dzu.float().cpu().numpy(), G.host(ggb), G.host(gbias))
    assert_close(out["fused"][1], out["sep"][1], rtol=1e-4, atol_scale=1e-5, what="dgamma/dbeta")
    assert_close_bf16(out["fused"][0], out["sep"][0], what="dz")


HEAD_CASES = [
    # n, ci, h, w
    (2, 64, 32, 32),
    (1, 64, 21, 37),     # ragged tiles
    (2, 32, 16, 16),
    (1, 100, 18, 16),    # C_in padded 100 -> 128
    (2, 128, 40, 24),    # two 64-channel chunks of the halo, several tiles per workgroup
]


@pytest.mark.parametrize("cfg", HEAD_CASES)
def test_one_channel_head_forward_dgrad_wgrad(G, capi, oracle, cfg):
    """The segmentation head's kernels (one output channel, fp32 [N][H][W] on the narrow side) against the
    CPU oracle's convolution: forward, data gradient (plain, accumulate) and weight + bias gradient."""
    n, ci, h, w = cfg
    cip = pad32(ci)
    assert capi.lib().unet_conv3x3_head_supported(cip) == 1
    rng = np.random.default_rng(ci + h)
    x = bf16_round(rng.standard_normal((n, ci, h, w)))
    wt = bf16_round(rng.standard_normal(ci * 9) * (1.0 / np.sqrt(9 * ci)))
    bias = np.array([0.37], np.float32)
    w_and_b = np.concatenate([wt, bias]).astype(np.float32)
    dz = rng.standard_normal((n, 1, h, w)).astype(np.float32)

    xt = to_nhwc(G, capi, x)
    fwd, _ = pack(G, capi, wt, ci, 1, 3)
    z = G.empty((n, 1, h, w))
    bd = G.dev(bias)
    capi.unet_conv3x3_head_bf16(G.p(xt), cip, G.p(fwd), G.p(bd), G.p(z), n, h, w, cip, 32, None)
    z_ref = oracle.convolve(x, w_and_b, 1, 3, 1, 1)
    assert_close(G.host(z), z_ref, rtol=1e-4, atol_scale=1e-5, what=f"head forward {cfg}")

    dx_ref, dw_ref = oracle.convolve_backward(x, dz, w_and_b, 3, 1, 1)
    dzd, wd = G.dev(dz), G.dev(wt)
    dx = torch.full((n, h, w, cip), float("nan"), dtype=torch.bfloat16, device=G.DEV)
    capi.unet_conv3x3_head_dgrad_bf16(G.p(dzd), G.p(wd), ci, G.p(dx), cip, None, 0, None, None, n, h, w, cip, 0, None)
    assert_close_bf16(from_nhwc(G, capi, dx, ci), dx_ref, what=f"head dgrad {cfg}")
    assert np.all(dx.float().cpu().numpy()[..., ci:] == 0), "padded gradient channels must be zero"
    capi.unet_conv3x3_head_dgrad_bf16(G.p(dzd), G.p(wd), ci, G.p(dx), cip, None, 0, None, None, n, h, w, cip, 1, None)
    assert_close_bf16(from_nhwc(G, capi, dx, ci), 2 * dx_ref, what=f"head dgrad accumulate {cfg}")

    gw, gb = G.empty((ci * 9,)), G.empty((1,))
    capi.unet_conv3x3_head_wgrad_bf16(G.p(dzd), G.p(xt), cip, G.p(gw), G.p(gb), n, h, w, cip, ci, None)
    assert_close(G.host(gw), dw_ref[:ci * 9], rtol=1e-4, atol_scale=1e-5, what=f"head wgrad {cfg}")
    assert_close(G.host(gb), dw_ref[ci * 9:], rtol=1e-4, atol_scale=1e-5, what=f"head bias gradient {cfg}")


def test_one_channel_head_dgrad_with_fused_bn_reduction(G, capi):
    """With bn_z the head's data-gradient kernel also emits the BN-backward partial sums of the layer below:
    same dx and same (dgamma, dbeta, dz) as the unfused pair of calls."""
    n, ci, h, w = 2, 64, 24, 40
    rng = np.random.default_rng(5)
    dz = G.dev(rng.standard_normal((n, 1, h, w)).astype(np.float32))
    wt = G.dev((rng.standard_normal(ci * 9) * 0.1).astype(np.float32))
    zp = torch.from_numpy(rng.standard_normal((n, h, w, ci)).astype(np.float32)).to(G.DEV).bfloat16()
    gbp = G.dev((rng.random(2 * ci) + 0.5).astype(np.float32))
    coef = G.dev(np.stack([rng.random(ci) + 0.5, rng.standard_normal(ci) * 0.3, rng.standard_normal(ci) * 0.1,
                           rng.random(ci) + 0.7], axis=1).astype(np.float32).reshape(-1))
    pixels, tiles = n * h * w, capi.lib().unet_conv3x3_stat_tiles(h, w)
    dx_a = torch.zeros((n, h, w, ci), dtype=torch.bfloat16, device=G.DEV)
    dx_b = torch.zeros_like(dx_a)
    parts = G.empty((n * tiles * ci * 2,))
    capi.unet_conv3x3_head_dgrad_bf16(G.p(dz), G.p(wt), ci, G.p(dx_a), ci, None, 0, None, None, n, h, w, ci, 0, None)
    capi.unet_conv3x3_head_dgrad_bf16(G.p(dz), G.p(wt), ci, G.p(dx_b), ci, G.p(zp), ci, G.p(coef), G.p(parts), n, h, w,
                                      ci, 0, None)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dx_a.float().cpu().numpy(), dx_b.float().cpu().numpy())
    out = {}
    for name in ("sep", "fused"):
        dzp = torch.zeros((n, h, w, ci), dtype=torch.bfloat16, device=G.DEV)
        ggb, gbias = G.empty((2 * ci,)), G.empty((ci,))
        if name == "fused":
            capi.unet_bn_act_backward_from_partials_bf16(G.p(dx_a), ci, G.p(zp), ci, G.p(coef), G.p(gbp), G.p(parts),
                                                         n * tiles, G.p(dzp), ci, G.p(ggb), G.p(gbias), ci, ci, pixels, None)
        else:
            capi.unet_bn_act_backward_bf16(G.p(dx_a), ci, None, 0, G.p(zp), ci, G.p(coef), G.p(gbp), G.p(dzp), ci, G.p(ggb),
                                           G.p(gbias), ci, ci, pixels, 0, None)
        torch.cuda.synchronize()
        out[name] = (dzp.float().cpu().numpy(), G.host(ggb))
    assert_close(out["fused"][1], out["sep"][1], rtol=1e-4, atol_scale=1e-5, what="dgamma/dbeta")
    assert_close_bf16(out["fused"][0], out["sep"][0], what="dz")


def test_first_layer_with_8_stored_channels(G, capi, oracle):
    """A 3-channel image stored as NHWC with 8 channels (16 bytes per pixel): the forward and weight-gradient
    entry points take CiP = 8 with weights packed for 32 and give what the 32-channel storage gives."""
    n, ci, co, h, w = 2, 3, 64, 37, 21
    rng = np.random.default_rng(3)
    x = bf16_round(rng.standard_normal((n, ci, h, w)))
    wt = bf16_round(rng.standard_normal(ci * co * 9) * 0.2)
    bias = (rng.standard_normal(co) * 0.5).astype(np.float32)
    dy = bf16_round(rng.standard_normal((n, co, h, w)))
    fwd, _ = pack(G, capi, wt, ci, co, 3)
    bd = G.dev(bias)
    x8, x32 = to_nhwc(G, capi, x, ld=8, cpad=8), to_nhwc(G, capi, x, ld=32, cpad=32)
    np.testing.assert_array_equal(x8.float().cpu().numpy()[..., :ci], x.transpose(0, 2, 3, 1))
    assert np.all(x8.float().cpu().numpy()[..., ci:] == 0)
    outs = []
    for xt, cp in ((x8, 8), (x32, 32)):
        out = torch.zeros((n, h, w, co), dtype=torch.bfloat16, device=G.DEV)
        capi.unet_conv3x3_bf16(G.p(xt), cp, G.p(fwd), G.p(bd), co, G.p(out), co, None, n, h, w, cp, co, 0, None)
        outs.append(from_nhwc(G, capi, out, co))
    np.testing.assert_array_equal(outs[0], outs[1])
    assert_close_bf16(outs[0], oracle.convolve(x, np.concatenate([wt, bias]).astype(np.float32), co, 3, 1, 1),
                      what="first layer forward")
    dyt = to_nhwc(G, capi, dy)
    gws = []
    for xt, cp in ((x8, 8), (x32, 32)):
        gw = G.empty((ci * co * 9,))
        capi.unet_conv3x3_wgrad_bf16(G.p(dyt), co, G.p(xt), cp, G.p(gw), n, h, w, cp, co, ci, co, None)
        gws.append(G.host(gw))
    np.testing.assert_array_equal(gws[0], gws[1])


@pytest.mark.parametrize("kind,shape", [("conv", (2, 64, 96, 20, 24)), ("tconv", (2, 64, 96, 20, 24)),
                                        ("conv", (2, 64, 128, 40, 24)),    # the pipelined kernel
                                        ("conv", (2, 64, 128, 24, 24))])   # its 16-row tiles
def test_inference_bn_relu_folded_into_the_epilogue(G, capi, oracle, kind, shape):
    """Inference: conv / transposed conv with BN (running statistics) + ReLU folded in, against the oracle's
    convolution -> bn_inference -> relu, and against the unfused pair of entry points."""
    n, ci, co, h, w = shape
    cip, cop = pad32(ci), pad32(co)
    k = 3 if kind == "conv" else 2
    rng = np.random.default_rng(31)
    x = bf16_round(rng.standard_normal((n, ci, h, w)))
    wt = bf16_round(rng.standard_normal(ci * co * k * k) * (1.0 / np.sqrt(k * k * ci)))
    bias = (rng.standard_normal(co) * 0.5).astype(np.float32)
    gb = np.stack([rng.random(co) + 0.5, rng.standard_normal(co) * 0.3], axis=1).astype(np.float32).reshape(-1)
    run = np.stack([rng.standard_normal(co) * 0.2, rng.random(co) + 0.5], axis=1).astype(np.float32).reshape(-1)
    w_and_b = np.concatenate([wt, bias]).astype(np.float32)
    xt = to_nhwc(G, capi, x)
    fwd, _ = pack(G, capi, wt, ci, co, k)
    bd, gbd, rund = G.dev(bias), G.dev(gb), G.dev(run)
    coef = G.empty((cop * 4,))
    capi.unet_bn_inference_coef_bf16(G.p(gbd), G.p(rund), co, cop, G.p(coef), None)
    ho, wo = (h, w) if kind == "conv" else (2 * h, 2 * w)
    y_fused = torch.full((n, ho, wo, cop), float("nan"), dtype=torch.bfloat16, device=G.DEV)
    z = torch.zeros((n, ho, wo, cop), dtype=torch.bfloat16, device=G.DEV)
    y_plain = torch.zeros_like(z)
    if kind == "conv":
        capi.unet_conv3x3_bn_relu_bf16(G.p(xt), cip, G.p(fwd), G.p(bd), co, G.p(coef), G.p(y_fused), cop, n, h, w, cip, cop, None)
        capi.unet_conv3x3_bf16(G.p(xt), cip, G.p(fwd), G.p(bd), co, G.p(z), cop, None, n, h, w, cip, cop, 0, None)
        z_ref = oracle.convolve(x, w_and_b, co, 3, 1, 1)
    else:
        capi.unet_tconv2x2_bn_relu_bf16(G.p(xt), cip, G.p(fwd), G.p(bd), co, G.p(coef), G.p(y_fused), cop, n, h, w, cip, cop, None)
        capi.unet_tconv2x2_bf16(G.p(xt), cip, G.p(fwd), G.p(bd), co, G.p(z), cop, None, n, h, w, cip, cop, None)
        z_ref = oracle.upsample(x, w_and_b, co, 2)
    capi.unet_bn_act_bf16(G.p(z), cop, G.p(y_plain), cop, G.p(coef), cop, n * ho * wo, 0, None)
    y_ref = oracle.relu(oracle.bn_inference(z_ref, gb, run))
    got = from_nhwc(G, capi, y_fused, co)
    assert_close_bf16(got, y_ref, what=f"{kind} + BN + ReLU folded")
    assert_close_bf16(got, from_nhwc(G, capi, y_plain, co), what=f"{kind}: folded vs unfused")
    assert np.all(y_fused.float().cpu().numpy()[..., co:] == 0), "padded channels must stay zero"


@pytest.mark.parametrize("kind", ["conv", "tconv"])
def test_fused_bn_reduction_is_bit_identical_between_runs(G, capi, kind):
    """The data-gradient kernels with the fused batch-norm reduction, repeated on identical inputs: dx and every
    partial sum must come out bit for bit the same.  The size matters: one build of the transposed-convolution
    kernel (its final products compiled to packed fp32 instructions) differed between runs only in workgroups
    that retire while others launch, i.e. only with more than one round of resident workgroups (768 at three per
    CU); 2048 workgroups here, and at that size about half of 60 repeats of that build differed."""
    n, cp, cn, h, w = 16, 128, 64, 128, 128
    rng = np.random.default_rng(5)
    k = 3 if kind == "conv" else 2
    hz, wz = (h, w) if kind == "conv" else (2 * h, 2 * w)
    dzn = torch.from_numpy(rng.standard_normal((n, hz, wz, cn)).astype(np.float32)).to(G.DEV).bfloat16()
    wflat = (rng.standard_normal(cp * cn * k * k + cn) * 0.1).astype(np.float32)
    _, wb = pack(G, capi, wflat, cp, cn, k)
    zp = torch.from_numpy(rng.standard_normal((n, h, w, cp)).astype(np.float32)).to(G.DEV).bfloat16()
    coef_h = np.stack([rng.random(cp) + 0.5, rng.standard_normal(cp) * 0.3, rng.standard_normal(cp) * 0.1,
                       rng.random(cp) + 0.7], axis=1).astype(np.float32)
    coef = G.dev(coef_h.reshape(-1))
    tiles = capi.lib().unet_conv3x3_stat_tiles(h, w)
    nrows = capi.lib().unet_conv3x3_stat_rows(n, h, w, cn, cp) if kind == "conv" else n * tiles
    first = None
    for _ in range(16):
        dx = torch.zeros((n, h, w, cp), dtype=torch.bfloat16, device=G.DEV)
        parts = G.empty((nrows * cp * 2,))
        if kind == "conv":
            capi.unet_conv3x3_dgrad_bnred_bf16(G.p(dzn), cn, G.p(wb), G.p(dx), cp, G.p(zp), cp, G.p(coef), G.p(parts), 0,
                                               n, h, w, cn, cp, None)
        else:
            capi.unet_tconv2x2_dgrad_bnred_bf16(G.p(dzn), cn, G.p(wb), G.p(dx), cp, G.p(zp), cp, G.p(coef), G.p(parts),
                                                n, h, w, cp, cn, None)
        torch.cuda.synchronize()
        if first is None:
            first = (dx.clone(), parts.clone())
        else:
            assert torch.equal(dx, first[0])
            assert torch.equal(parts, first[1])
