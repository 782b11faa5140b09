"""Shared comparison helpers.

Tolerances (the bar stated for this path): fp32 relative <= 1e-4, bf16 relative
<= 2e-2, integer/byte/index/layout results bit-exact.  "Relative" is evaluated
as |a-b| <= rtol*max(|a|,|b|) + atol with atol = 1e-6 * max(1, max|ref|): the
1e-6 absolute floor is the one the reference's own isClose uses
(src/cuda_library/test/test.h:10-17), scaled by the tensor's magnitude so that
near-zero elements of a long fp32 reduction are judged against the size of the
terms that were summed.
"""
import numpy as np

FP32_RTOL = 1e-4
BF16_RTOL = 2e-2


def assert_close(a, ref, rtol=FP32_RTOL, atol_scale=1e-6, what=""):
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, f"{what}: shape {a.shape} vs {ref.shape}"
    assert np.isfinite(a).all(), f"{what}: non-finite values in result"
    scale = max(1.0, float(np.max(np.abs(ref))) if ref.size else 1.0)
    atol = atol_scale * scale
    err = np.abs(a - ref)
    bound = rtol * np.maximum(np.abs(a), np.abs(ref)) + atol
    bad = err > bound
    if bad.any():
        i = np.unravel_index(np.argmax(err - bound), err.shape)
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{a.size} elements outside rtol={rtol} atol={atol:.3g}; "
            f"worst at {i}: got {a[i]!r} want {ref[i]!r}")


def assert_close_bf16(a, ref, what=""):
    """bf16 storage: 2e-2 relative, with an absolute floor of 2e-2 x rms(ref)
    (a bf16 value carries 8 significant bits, so elements much smaller than the
    tensor's typical magnitude are judged against that magnitude)."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, f"{what}: shape {a.shape} vs {ref.shape}"
    assert np.isfinite(a).all(), f"{what}: non-finite values in result"
    rms = float(np.sqrt(np.mean(ref * ref))) if ref.size else 0.0
    err = np.abs(a - ref)
    bound = BF16_RTOL * np.maximum(np.abs(a), np.abs(ref)) + BF16_RTOL * rms
    bad = err > bound
    if bad.any():
        i = np.unravel_index(np.argmax(err - bound), err.shape)
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{a.size} elements outside bf16 tolerance; "
            f"worst at {i}: got {a[i]!r} want {ref[i]!r} (rms {rms:.3g})")


def assert_close_bf16_net(a, ref, what="", rel_bar=None, guard=True):
    """Activations / gradients that passed through several bf16 layers: each
    layer adds rounding noise of ~0.5 % of the tensor's rms (bf16 keeps 8
    significant bits; a 3x3 convolution sums hundreds of rounded products), so
    the bound is on the relative L2 error (<= 2e-2) with a per-element guard of
    2e-2*|x| + 0.15*rms against outliers (the tail of the accumulated rounding noise
    reaches several sigma among ~1e5 elements)."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, f"{what}: shape {a.shape} vs {ref.shape}"
    assert np.isfinite(a).all(), f"{what}: non-finite values in result"
    rms = float(np.sqrt(np.mean(ref * ref))) if ref.size else 0.0
    rel = float(np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-30))
    rel_bar = BF16_RTOL if rel_bar is None else rel_bar
    assert rel <= rel_bar, f"{what}: relative L2 error {rel:.3g} > {rel_bar:.3g}"
    if not guard:
        return
    err = np.abs(a - ref)
    bound = BF16_RTOL * np.maximum(np.abs(a), np.abs(ref)) + 0.15 * rms
    bad = err > bound
    if bad.any():
        i = np.unravel_index(np.argmax(err - bound), err.shape)
        raise AssertionError(f"{what}: {int(bad.sum())}/{a.size} elements outside the per-element guard; "
                             f"worst at {i}: got {a[i]!r} want {ref[i]!r} (rms {rms:.3g})")


def assert_exact(a, ref, what=""):
    a = np.asarray(a)
    ref = np.asarray(ref)
    assert a.shape == ref.shape, f"{what}: shape {a.shape} vs {ref.shape}"
    if not np.array_equal(a, ref):
        bad = np.argwhere(a != ref)
        raise AssertionError(f"{what}: {len(bad)} mismatches, first at {tuple(bad[0])}: "
                             f"{a[tuple(bad[0])]!r} vs {ref[tuple(bad[0])]!r}")
