"""Operators beside a busy second stream: every result must be bit-identical to the one computed alone.

One process runs several streams (the data-parallel driver's communication stream, a caller's own), so waves of
different kernels share CUs and SIMDs.  One build of the pooled batch-norm backward returned wrong channel sums
only then -- its packed-fp32 arithmetic went wrong while waves of the 3x3 weight-gradient kernel (MFMA, 248
registers) ran on the same SIMD, with every input intact -- which no single-stream test can see.  The co-runner
here is that weight-gradient kernel, launched repeatedly on a second stream around each call."""
import numpy as np
import pytest
import torch

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


def _bf(G, rng, shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32)).to(G.DEV).bfloat16()


def _coef(G, capi, rng, c):
    full = torch.zeros(capi.lib().unet_bn_coef_floats(c), device=G.DEV)
    rec = np.stack([rng.random(c) + 0.5, rng.standard_normal(c) * 0.3, rng.standard_normal(c) * 0.1, rng.random(c) + 0.7], 1)
    full[:4 * c] = torch.from_numpy(rec.astype(np.float32).reshape(-1)).to(G.DEV)
    return full


def _victims(G, capi):
    """name -> callable returning the op's output tensors (inputs are created once, outputs per call)."""
    rng = np.random.default_rng(7)
    n, h, w, c = 16, 32, 32, 256
    dy, z = _bf(G, rng, (n, h, w, c)), _bf(G, rng, (n, h, w, c))
    pooled = _bf(G, rng, (n, h // 2, w // 2, c))
    mask = torch.from_numpy(rng.integers(1, 16, (n, h // 2, w // 2, c)).astype(np.uint8)).to(G.DEV)
    coef = _coef(G, capi, rng, c)
    gb = torch.from_numpy(np.stack([rng.random(c) + 0.5, rng.standard_normal(c)], 1).astype(np.float32).reshape(-1)).to(G.DEV)
    ci, co = 128, 128
    x = _bf(G, rng, (n, h, w, ci))
    wflat = torch.from_numpy((rng.standard_normal(ci * co * 9 + co) * 0.05).astype(np.float32)).to(G.DEV)
    wf = torch.zeros(9 * co * ci, dtype=torch.bfloat16, device=G.DEV)
    wb = torch.zeros(9 * co * ci, dtype=torch.bfloat16, device=G.DEV)
    capi.unet_pack_weights_bf16(G.p(wflat), G.p(wf), G.p(wb), ci, co, ci, co, 3, None)
    bias = wflat[ci * co * 9:].clone()
    zp, coefp = _bf(G, rng, (n, h, w, ci)), _coef(G, capi, rng, ci)
    rows = capi.lib().unet_conv3x3_stat_rows(n, h, w, ci, co)
    tw = torch.from_numpy((rng.standard_normal(ci * 64 * 4 + 64) * 0.05).astype(np.float32)).to(G.DEV)
    twf = torch.zeros(4 * 64 * ci, dtype=torch.bfloat16, device=G.DEV)
    twb = torch.zeros(4 * 64 * ci, dtype=torch.bfloat16, device=G.DEV)
    capi.unet_pack_weights_bf16(G.p(tw), G.p(twf), G.p(twb), ci, 64, ci, 64, 2, None)
    dyt = _bf(G, rng, (n, 2 * h, 2 * w, 64))
    tiles = capi.lib().unet_conv3x3_stat_tiles(h, w)

    def bn_bwd_pooled():
        dz, ggb = torch.zeros_like(z), torch.zeros(2 * c, device=G.DEV)
        capi.unet_bn_act_backward_pooled_bf16(G.p(dy), c, G.p(pooled), c, G.p(mask), n, h, w, G.p(z), c, G.p(coef), G.p(gb),
                                              G.p(dz), c, G.p(ggb), None, c, c, 0, None)
        return dz.view(torch.int16), ggb

    def bn_bwd():
        dz, ggb = torch.zeros_like(z), torch.zeros(2 * c, device=G.DEV)
        capi.unet_bn_act_backward_bf16(G.p(dy), c, None, 0, G.p(z), c, G.p(coef), G.p(gb), G.p(dz), c, G.p(ggb), None, c, c,
                                       n * h * w, 0, None)
        return dz.view(torch.int16), ggb

    def bn_act_pool():
        y, pl = torch.zeros_like(z), torch.zeros_like(pooled)
        mk = torch.zeros_like(mask)
        capi.unet_bn_act_pool_bf16(G.p(z), c, G.p(y), c, G.p(coef), G.p(pl), c, G.p(mk), n, h, w, c, 0, None)
        return y.view(torch.int16), pl.view(torch.int16), mk

    def conv_fwd_stats():
        out = torch.zeros((n, h, w, co), dtype=torch.bfloat16, device=G.DEV)
        st = torch.zeros(rows * co * 2, device=G.DEV)
        capi.unet_conv3x3_bf16(G.p(x), ci, G.p(wf), G.p(bias), co, G.p(out), co, G.p(st), n, h, w, ci, co, 0, None)
        return out.view(torch.int16), st

    def conv_dgrad_bnred():
        dx = torch.zeros((n, h, w, ci), dtype=torch.bfloat16, device=G.DEV)
        parts = torch.zeros(capi.lib().unet_conv3x3_stat_rows(n, h, w, co, ci) * ci * 2, device=G.DEV)
        capi.unet_conv3x3_dgrad_bnred_bf16(G.p(dy[..., :co].contiguous()), co, G.p(wb), G.p(dx), ci, G.p(zp), ci, G.p(coefp),
                                           G.p(parts), 0, n, h, w, co, ci, None)
        return dx.view(torch.int16), parts

    def tconv_dgrad_bnred():
        dx = torch.zeros((n, h, w, ci), dtype=torch.bfloat16, device=G.DEV)
        parts = torch.zeros(n * tiles * ci * 2, device=G.DEV)
        capi.unet_tconv2x2_dgrad_bnred_bf16(G.p(dyt), 64, G.p(twb), G.p(dx), ci, G.p(zp), ci, G.p(coefp), G.p(parts), n, h, w,
                                            ci, 64, None)
        return dx.view(torch.int16), parts

    return {"bn_bwd_pooled": bn_bwd_pooled, "bn_bwd": bn_bwd, "bn_act_pool": bn_act_pool, "conv_fwd_stats": conv_fwd_stats,
            "conv_dgrad_bnred": conv_dgrad_bnred, "tconv_dgrad_bnred": tconv_dgrad_bnred}


def test_results_do_not_depend_on_a_busy_second_stream(G, capi):
    victims = _victims(G, capi)
    rng = np.random.default_rng(8)
    n, h, w, ci, co = 16, 16, 16, 512, 1024            # the bottleneck's weight gradient: 150 us of MFMA per call
    xw, dzw = _bf(G, rng, (n, h, w, ci)), _bf(G, rng, (n, h, w, co))
    gw = torch.zeros(ci * co * 9, device=G.DEV)
    side = torch.cuda.Stream()
    failures = []
    for name, fn in victims.items():
        torch.cuda.synchronize()
        alone = [t.clone() for t in fn()]
        torch.cuda.synchronize()
        for rep in range(6):
            for _ in range(3):
                capi.unet_conv3x3_wgrad_bf16(G.p(dzw), co, G.p(xw), ci, G.p(gw), n, h, w, ci, co, ci, co, side.cuda_stream)
            got = fn()
            torch.cuda.synchronize()
            if not all(torch.equal(a, b) for a, b in zip(alone, got)):
                failures.append(f"{name} (repeat {rep})")
                break
    assert not failures, "results changed beside a concurrent kernel: " + ", ".join(failures)
