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


class _FakeNet:
    """The slice of neural_net.NeuralNet that save()/load() use, on host arrays."""

    def __init__(self, seed):
        rng = np.random.default_rng(seed)
        self.layout = [(0, 0, 6, 4), (2, 10, 4, 6)]
        self.params = rng.standard_normal(20).astype(np.float32)
        self.running = {0: rng.standard_normal(4).astype(np.float32), 2: rng.standard_normal(6).astype(np.float32)}
        self.adam = rng.standard_normal(40).astype(np.float32)
        self.steps = 7
        self.writes = 0

    def parameter_layout(self): return self.layout
    def num_parameters(self): return self.params.size
    def get_parameters(self): return self.params.copy()
    def running_stats(self, i): return self.running[i].copy()
    def get_optimizer_state(self): return self.adam.copy()
    def optimizer_steps(self): return self.steps

    def set_parameters(self, p):
        self.writes += 1
        self.params = np.array(p)

    def set_running_stats(self, i, r):
        self.writes += 1
        self.running[i] = np.array(r)

    def set_optimizer_state(self, a, steps):
        self.writes += 1
        self.adam, self.steps = np.array(a), steps


def test_save_load_agree_on_the_file_name_and_validate_before_writing(tmp_path):
    import pytest
    a, b = _FakeNet(1), _FakeNet(2)
    for name in ("ckpt", "ckpt2.npz"):                      # with and without the suffix numpy appends
        ck.save(a, tmp_path / name)
        ck.load(b, tmp_path / name)
        np.testing.assert_array_equal(b.params, a.params)
        np.testing.assert_array_equal(b.running[2], a.running[2])
        np.testing.assert_array_equal(b.adam, a.adam)
        assert b.steps == 7
    assert (tmp_path / "ckpt.npz").exists() and not (tmp_path / "ckpt2.npz.npz").exists()
    # a file whose arrays do not fit the network is rejected before anything is copied in
    with np.load(tmp_path / "ckpt.npz") as f:
        arrays = {k: f[k] for k in f.files}
    for key, bad in (("parameters", arrays["parameters"][:-1]), ("running_2", arrays["running_2"][:-2]),
                     ("adam_state", arrays["adam_state"][:-3])):
        broken = dict(arrays)
        broken[key] = bad
        np.savez(tmp_path / "broken.npz", **broken)
        c = _FakeNet(3)
        with pytest.raises(RuntimeError, match="checkpoint"):
            ck.load(c, tmp_path / "broken")
        assert c.writes == 0
