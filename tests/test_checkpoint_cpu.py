"""Layout conversions of unet_amd.checkpoint (no GPU): reference [C_in][C_out][k][k] <-> PyTorch."""
import numpy as np
import torch
import torch.nn.functional as F

from unet_amd import checkpoint as ck


def test_conv_weights_round_trip_and_match_torch(oracle):
    rng = np.random.default_rng(0)
    ci, co, k = 3, 5, 3
    wb = rng.standard_normal(ci * co * k * k + co).astype(np.float32)
    w, b = ck.layer_to_torch(wb, ci, co, k, transposed=False)
    assert w.shape == (co, ci, k, k)
    np.testing.assert_array_equal(ck.layer_from_torch(w, b, False), wb)
    x = rng.standard_normal((2, ci, 6, 7)).astype(np.float32)
    want = oracle.convolve(x, wb, co, k, 1, 1)
    got = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_transposed_conv_weights_round_trip_and_match_torch(oracle):
    rng = np.random.default_rng(1)
    ci, co, k = 4, 3, 2
    wb = rng.standard_normal(ci * co * k * k + co).astype(np.float32)
    w, b = ck.layer_to_torch(wb, ci, co, k, transposed=True)
    assert w.shape == (ci, co, k, k)
    np.testing.assert_array_equal(ck.layer_from_torch(w, b, True), wb)
    x = rng.standard_normal((2, ci, 5, 4)).astype(np.float32)
    want = oracle.upsample(x, wb, co, k)
    got = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=k).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_split_parameters_deinterleaves_bn():
    flat = np.arange(20, dtype=np.float32)
    parts = ck.split_parameters(flat, [(0, 0, 6, 4), (2, 10, 4, 6)])
    np.testing.assert_array_equal(parts[0]["wb"], np.arange(6))
    np.testing.assert_array_equal(parts[0]["gamma"], [6, 8])
    np.testing.assert_array_equal(parts[0]["beta"], [7, 9])
    np.testing.assert_array_equal(parts[2]["gamma"], [14, 16, 18])
