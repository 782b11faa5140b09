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


def test_attention_gate_parameters_round_trip_and_match_torch(oracle):
    """attention_to_torch() tensors drive stock torch operators (Conv2d 1x1 stride 2 without bias on x, Conv2d 1x1
    on g, Conv2d 1x1 to one channel) to the oracle's intermediate tensors; from_torch inverts it exactly."""
    import torch
    import torch.nn.functional as F
    from unet_amd import checkpoint
    rng = np.random.default_rng(5)
    n, cx, cg, ci, h, w = 2, 6, 10, 3, 8, 12
    flat = (rng.standard_normal(oracle.attention_params(cx, cg, ci)) * 0.3).astype(np.float32)
    t = checkpoint.attention_to_torch(flat, cx, cg)
    assert set(t) == set(checkpoint.ATTENTION_KEYS)
    np.testing.assert_array_equal(checkpoint.attention_from_torch(t), flat)
    x = rng.standard_normal((n, cx, h, w)).astype(np.float32)
    g = rng.standard_normal((n, cg, h // 2, w // 2)).astype(np.float32)
    _, int1, alpha = oracle.attention(x, g, flat, ci)
    tt = {k: torch.from_numpy(v).double() for k, v in t.items()}
    q = torch.relu(F.conv2d(torch.from_numpy(x).double(), tt["w_x.weight"], stride=2) +
                   F.conv2d(torch.from_numpy(g).double(), tt["w_g.weight"], tt["w_g.bias"]))
    a = torch.sigmoid(F.conv2d(q, tt["psi.weight"], tt["psi.bias"]))[:, 0]
    np.testing.assert_allclose(q.numpy(), int1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a.numpy(), alpha, rtol=1e-5, atol=1e-6)
    with np.testing.assert_raises(RuntimeError):
        checkpoint.attention_to_torch(flat[:-1], cx, cg)
