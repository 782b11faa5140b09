"""GPU tests of the bf16 / NHWC compute path of `neural_net.NeuralNet` (the
default path when every channel count is a multiple of 32): one training step
against PyTorch-CPU float64 on identical parameters and inputs.

Two oracles:
  * float64 PyTorch that rounds to bf16 exactly where the module stores bf16
    (torch_ref.forward(emulate_bf16=True)), so ReLU masks and pool ties agree
    except where an activation sits within one bf16 ulp of a decision point:
    every layer's activation within relative 2e-2 (measured: <= 7e-3), loss
    within 2e-3;
  * plain float64 PyTorch: the loss within 2e-2; activations within
    max(2e-2, 0.8e-2*sqrt(d)) relative L2 after d bf16 conv layers (each
    layer's storage rounding adds ~0.6 % of independent noise, so the 2e-2 bar
    holds up to d = 6 and the noise reaches ~2.4e-2 at d = 9).

Gradients: every backward OPERATOR meets the 2e-2 bar on identical inputs
(tests/test_ops_bf16_gpu.py), and so does a shallow network here.  Through a
deep network a per-element bar is not meaningful for any bf16 implementation:
activation differences of one bf16 ulp (0.4 %) flip the ReLU mask of ~0.3 % of
the elements of each layer, every flip changes one gradient term by its full
size, and the relative L2 difference therefore grows like
sqrt(layers x flipped fraction): ~0.05 per layer against the bf16-storage
oracle, ~0.1 per layer against plain float64.  That difference is zero-mean
noise, so deep-network gradients are held to a cosine similarity (>= 0.99 vs
the bf16-storage oracle, >= 0.9 vs float64) and a relative L2 error <= 0.15
against the bf16-storage oracle.
"""
import os

import numpy as np
import pytest
import torch

import torch_ref
from util import assert_close_bf16_net, assert_exact

pytestmark = pytest.mark.gpu
os.environ.setdefault("UNET_QUIET", "1")


@pytest.fixture(scope="module")
def nn():
    import neural_net
    return neural_net


def _make(nn, graph, loss=None, seed=3, dtype=None):
    old = os.environ.get("UNET_DTYPE")
    if dtype:
        os.environ["UNET_DTYPE"] = dtype
    try:
        net = nn.NeuralNet()
        net.set_seed(seed)
        net.create(torch_ref.to_layer_descs(nn, graph), loss if loss is not None else nn.LossFunction.DICE_LOSS)
    finally:
        if dtype:
            if old is None:
                del os.environ["UNET_DTYPE"]
            else:
                os.environ["UNET_DTYPE"] = old
    return net


def _with_acts(graph, hidden):
    """The graph with its hidden convolutions' activations cycling through `hidden` (the last layer keeps its own)."""
    out, k = [], 0
    for i, (kind, s, parents, act) in enumerate(graph):
        if kind == "conv" and i != len(graph) - 1:
            act = hidden[k % len(hidden)]
            k += 1
        out.append((kind, s, parents, act))
    return out


GRAPHS = {
    # conv(3->32)+BN+ReLU -> conv(32->1)+BN+sigmoid: one hidden ReLU layer
    "shallow": lambda: [("conv", [32, 48, 3, 32, 3, 1, 1], [], "relu"), ("conv", [32, 48, 32, 1, 3, 1, 1], [0], "sigmoid")],
    "mainpy32": lambda: torch_ref.main_py_graph(32, 48, 3, 32),
    "unet2_double": lambda: torch_ref.unet_graph(32, 32, 3, 32, 2, True),
    # LEAKY_ReLU (with its slope) and TANH on the hidden convolutions: the fused bf16 batch-norm kernels' act 3 and 4
    "mainpy32_leaky_tanh": lambda: _with_acts(torch_ref.main_py_graph(32, 48, 3, 32), ["leaky:0.1", "tanh", "relu"]),
}


def test_path_selection(nn):
    g = torch_ref.main_py_graph(32, 32, 3, 32)
    assert _make(nn, g).compute_dtype() == "bf16"
    assert _make(nn, g, dtype="fp32").compute_dtype() == "f32"
    assert _make(nn, torch_ref.main_py_graph(16, 16, 3, 4)).compute_dtype() == "f32"   # channels off multiples of 32


@pytest.mark.parametrize("name", list(GRAPHS))
@pytest.mark.parametrize("loss_name", ["dice", "difference"])
def test_train_step_bf16_matches_torch(nn, name, loss_name):
    graph = GRAPHS[name]()
    loss = nn.LossFunction.DICE_LOSS if loss_name == "dice" else nn.LossFunction.DIFFERENCE_LOSS
    net = _make(nn, graph, loss)
    assert net.compute_dtype() == "bf16"
    h, w = graph[0][1][0], graph[0][1][1]
    rng = np.random.default_rng(7)
    n = 4
    x = rng.standard_normal((n, 3, h, w)).astype(np.float32)
    m = (rng.random((n, h, w)) < 0.3).astype(np.float32)
    p0 = net.get_parameters()
    p0 = (p0 + rng.standard_normal(p0.size).astype(np.float32) * 0.02).astype(np.float32)
    net.set_parameters(p0)
    loss_val = net.train_step(x, m, 1e-2, False)
    grads = net.get_gradients()

    loss_fn = torch_ref.dice_loss if loss_name == "dice" else torch_ref.difference_loss
    outs = {}
    for emu in (False, True):
        params = torch_ref.split_params(graph, p0)
        acts = torch_ref.forward(graph, params, torch.from_numpy(x).double(), emulate_bf16=emu)
        ref_loss = loss_fn(acts[-1], torch.from_numpy(m).double())
        ref_loss.backward()
        outs[emu] = ([a.detach().numpy() for a in acts], float(ref_loss.detach()), torch_ref.flat_grads(graph, params))

    acts, ref_loss, plain_grads = outs[False]
    emu_acts, emu_loss, ref_grads = outs[True]
    depth = 0
    for i in range(len(graph)):
        depth += graph[i][0] in ("conv", "up")
        got = net.layer_output(i, n)
        assert_close_bf16_net(got, emu_acts[i], what=f"{name} layer {i} activation vs bf16-storage oracle")
        assert_close_bf16_net(got, acts[i], what=f"{name} layer {i} activation vs float64 oracle",
                              rel_bar=max(2e-2, 0.8e-2 * np.sqrt(depth)), guard=False)
    assert abs(loss_val - ref_loss) <= 2e-2 * abs(ref_loss) + 1e-6
    assert abs(loss_val - emu_loss) <= 2e-3 * abs(emu_loss) + 1e-6
    worst = 0.0
    for layer, off, nwb, nbn in net.parameter_layout():
        co = nbn // 2
        for lo, hi, what in ((off, off + nwb - co, "weights"), (off + nwb, off + nwb + nbn, "gamma/beta")):
            g, r, pl = grads[lo:hi].astype(np.float64), ref_grads[lo:hi], plain_grads[lo:hi]
            rel = np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-12)
            worst = max(worst, rel)
            bar = 2e-2 if name == "shallow" else 0.15
            assert rel <= bar, f"{name} layer {layer} {what} gradient vs bf16-storage oracle: relative L2 error {rel:.3g}"
            cos_e = float(np.dot(g, r) / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
            cos_p = float(np.dot(g, pl) / max(np.linalg.norm(g) * np.linalg.norm(pl), 1e-30))
            assert cos_e >= 0.99, f"{name} layer {layer} {what} gradient vs bf16-storage oracle: cosine {cos_e:.4f}"
            assert cos_p >= 0.9, f"{name} layer {layer} {what} gradient vs float64 oracle: cosine {cos_p:.4f}"
        # the conv bias feeds a batch norm, so its true gradient is zero: only rounding noise may remain
        gb = grads[off + nwb - co:off + nwb]
        wmax = np.abs(grads[off:off + nwb - co]).max()
        assert np.abs(gb).max() <= 0.05 * max(wmax, 1e-6) * np.sqrt(x.size / 3), f"layer {layer} bias gradient {gb}"
    print(f"{name}/{loss_name}: worst gradient relative L2 error {worst:.3g}")


def test_bf16_training_learns_and_infers(nn):
    graph = torch_ref.main_py_graph(32, 32, 3, 32)
    net = _make(nn, graph)
    rng = np.random.default_rng(2)
    n = 10
    x = rng.standard_normal((n, 3, 32, 32)).astype(np.float32)
    m = (x[:, 0] > 0.3).astype(np.float32)
    net.train(x, m, 3e-3, 1, 4)
    first = net.last_epoch_loss()
    net.train(x, m, 3e-3, 15, 4)
    last = net.last_epoch_loss()
    assert np.isfinite(first) and np.isfinite(last) and last < first - 0.05, (first, last)
    out = net.infer(x[0])
    assert out.shape == (1, 32, 32) and set(np.unique(out)) <= {0.0, 1.0}
    d = net.test(x[0], (m[0] > 0).astype(np.uint8))
    assert 0.0 <= d <= 1.0
    # inference forward (running statistics) agrees with PyTorch-CPU on the trained parameters
    params = torch_ref.split_params(graph, net.get_parameters())
    running = [net.running_stats(i) for i in range(len(graph))]
    acts = torch_ref.forward(graph, params, torch.from_numpy(x[:2]).double(), training=False, running=running)
    assert_close_bf16_net(net.forward(x[:2], True), acts[-1].detach().numpy(), what="bf16 inference forward")


def test_bf16_deterministic(nn):
    graph = torch_ref.main_py_graph(32, 32, 3, 32)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((4, 3, 32, 32)).astype(np.float32)
    m = (rng.random((4, 32, 32)) < 0.3).astype(np.float32)
    outs = []
    for _ in range(2):
        net = _make(nn, graph, seed=11)
        net.train_step(x, m, 1e-2, True)
        net.train_step(x, m, 1e-2, True)
        outs.append((net.get_parameters(), net.get_gradients()))
    assert_exact(outs[0][0], outs[1][0], what="parameters bitwise reproducible")
    assert_exact(outs[0][1], outs[1][1], what="gradients bitwise reproducible")
