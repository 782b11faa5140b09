"""The entry point that allocates nothing (unet_batch_norm_train_noalloc_f32) against the allocating one
(unet_batch_norm_train_f32): same kernels, so every output must be bit-identical; two calls in a row,
the second with a smaller input, so stale scratch contents would show."""
import numpy as np
import pytest
import torch

from unet_amd import capi
from gpu_util import dev, empty, host, p
from util import assert_close

pytestmark = pytest.mark.gpu


def _run(fn, x, gb, running0):
    n, c, h, w = x.shape
    y = empty(x.shape)
    stats = empty((2 * c,))
    running = running0.clone()
    fn(p(x), p(y), p(gb), p(stats), p(running), c, h, w, n, None)
    return host(y), host(stats), host(running)


def test_noalloc_batch_norm_train_matches_allocating_entry_point(oracle):
    rng = np.random.default_rng(7)
    for shape in [(4, 48, 96, 80), (2, 5, 17, 13)]:  # the second call is the smaller one
        n, c, h, w = shape
        xh = rng.standard_normal(shape).astype(np.float32) * 1.5 + 0.3
        gbh = rng.standard_normal(2 * c).astype(np.float32)
        rh = np.abs(rng.standard_normal(2 * c)).astype(np.float32)
        x, gb, r0 = dev(xh), dev(gbh), dev(rh)
        ref = _run(capi.unet_batch_norm_train_f32, x, gb, r0)
        got = _run(capi.unet_batch_norm_train_noalloc_f32, x, gb, r0)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)
        # and against the CPU oracle, to the fp32 tolerance of the other batch-norm tests
        y_o, stats_o, run_o = oracle.bn_train(xh, gbh, rh)
        assert_close(got[0], y_o, what="bn y")
        assert_close(got[1], stats_o, what="bn raw sums")
        assert_close(got[2], run_o, what="bn running stats")


def test_noalloc_batch_norm_train_rejects_oversize():
    c = 1024  # 1024 channels x 512 slices > the fixed scratch
    n, h, w = 8, 1024, 1024
    x = torch.empty(1, device="cuda:0")
    assert 2 * c * 512 > 2 * capi.lib().unet_batch_norm_train_noalloc_max_partials()
    with pytest.raises(RuntimeError, match="scratch"):
        capi.unet_batch_norm_train_noalloc_f32(p(x), p(x), p(x), p(x), None, c, h, w, n, None)
