"""Checkpoint save/load and PyTorch weight exchange on the GPU module."""
import os

import numpy as np
import pytest
import torch

import torch_ref
from unet_amd import checkpoint as ck

pytestmark = pytest.mark.gpu
os.environ.setdefault("UNET_QUIET", "1")


def _net(nn, graph, seed):
    net = nn.NeuralNet()
    net.set_seed(seed)
    descs = torch_ref.to_layer_descs(nn, graph)
    net.create(descs, nn.LossFunction.DICE_LOSS)
    return net, descs


def test_save_load_resumes_bit_identically(tmp_path):
    import neural_net as nn
    graph = torch_ref.unet_graph(32, 32, 3, 32, 2, True)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((4, 3, 32, 32)).astype(np.float32)
    m = (rng.random((4, 32, 32)) < 0.3).astype(np.float32)
    a, _ = _net(nn, graph, 1)
    for _ in range(3):
        a.train_step(x, m, 1e-2, True)
    path = str(tmp_path / "ck.npz")
    ck.save(a, path)
    b, _ = _net(nn, graph, 2)            # different initial weights
    ck.load(b, path)
    assert np.array_equal(a.get_parameters(), b.get_parameters())
    assert np.array_equal(a.forward(x, True), b.forward(x, True))      # running statistics restored
    la, lb = a.train_step(x, m, 1e-2, True), b.train_step(x, m, 1e-2, True)
    assert la == lb and np.array_equal(a.get_parameters(), b.get_parameters())   # Adam state and step count restored
    c, _ = _net(nn, torch_ref.main_py_graph(32, 32, 3, 32), 1)
    with pytest.raises(RuntimeError, match="different layer list"):
        ck.load(c, path)


def test_weights_exchange_with_pytorch_modules():
    import neural_net as nn
    graph = torch_ref.main_py_graph(16, 16, 3, 4)      # fp32 path (channels off multiples of 32)
    net, descs = _net(nn, graph, 5)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 3, 16, 16)).astype(np.float32)
    m = (rng.random((2, 16, 16)) < 0.3).astype(np.float32)
    net.train_step(x, m, 1e-2, True)
    state = ck.to_torch(net, descs)
    # a PyTorch restatement of the graph, loaded through load_state_dict-style tensors
    acts = []
    for i, (kind, s, parents, act) in enumerate(graph):
        inp = torch.from_numpy(x) if not parents else acts[parents[0]]
        if kind in ("conv", "up"):
            w, b = torch.from_numpy(state[f"layers.{i}.conv.weight"]), torch.from_numpy(state[f"layers.{i}.conv.bias"])
            z = (torch.nn.functional.conv2d(inp, w, b, padding=s[5], stride=s[6]) if kind == "conv"
                 else torch.nn.functional.conv_transpose2d(inp, w, b, stride=s[4]))
            bn = torch.nn.BatchNorm2d(s[3], eps=1e-8)
            bn.load_state_dict({"weight": torch.from_numpy(state[f"layers.{i}.bn.weight"]),
                                "bias": torch.from_numpy(state[f"layers.{i}.bn.bias"]),
                                "running_mean": torch.from_numpy(state[f"layers.{i}.bn.running_mean"]),
                                "running_var": torch.from_numpy(state[f"layers.{i}.bn.running_var"]),
                                "num_batches_tracked": torch.tensor(1)})
            bn.eval()
            y = bn(z)
            acts.append(torch.sigmoid(y) if act == "sigmoid" else torch.relu(y))
        elif kind == "pool":
            acts.append(torch.nn.functional.max_pool2d(inp, s[3], s[4]))
        else:
            acts.append(torch.cat([acts[parents[0]], acts[parents[1]]], dim=1))
    want = acts[-1].detach().numpy()
    got = net.forward(x, True)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5)
    # and back: a fresh network takes the PyTorch tensors
    other, _ = _net(nn, graph, 9)
    ck.from_torch(other, descs, {k: torch.from_numpy(v) for k, v in state.items()})
    assert np.array_equal(other.get_parameters(), net.get_parameters())
    assert np.array_equal(other.forward(x, True), got)


def test_weights_exchange_covers_bn_and_fc_layers():
    """A BN_LAYER travels as BatchNorm2d tensors only, an FC_LAYER as torch.nn.Linear's [out, in] weight and bias."""
    import neural_net as nn
    L, T, A = nn.LayerDesc, nn.LayerType, nn.ActivationType
    descs = [L(T.CONV_LAYER, [8, 8, 3, 4, 3, 1, 1], [], A.SIGMOID), L(T.BN_LAYER, [], [0], A.TANH),
             L(T.FC_LAYER, [4, 8, 8, 3], [1], A.SOFTMAX)]
    nets = []
    for seed in (3, 8):
        net = nn.NeuralNet()
        net.set_seed(seed)
        net.create(descs, nn.LossFunction.CROSS_ENTROPY_LOSS)
        nets.append(net)
    a, b = nets
    rng = np.random.default_rng(2)
    x = rng.standard_normal((4, 3, 8, 8)).astype(np.float32)
    a.train_step(x, rng.integers(0, 3, (4, 1, 1)).astype(np.float32), 1e-2, True)   # moves the running statistics
    state = ck.to_torch(a, descs)
    assert state["layers.2.fc.weight"].shape == (3, 256) and state["layers.2.fc.bias"].shape == (3,)
    assert "layers.1.conv.weight" not in state and state["layers.1.bn.running_var"].shape == (4,)
    lin = torch.nn.Linear(256, 3)
    lin.load_state_dict({"weight": torch.from_numpy(state["layers.2.fc.weight"]), "bias": torch.from_numpy(state["layers.2.fc.bias"])})
    ck.from_torch(b, descs, {k: torch.from_numpy(v) for k, v in state.items()})
    assert np.array_equal(a.get_parameters(), b.get_parameters())
    got = a.forward(x, True)
    assert np.array_equal(got, b.forward(x, True))
    hidden = torch.from_numpy(a.layer_output(1, 4)).reshape(4, 256)
    np.testing.assert_allclose(got.reshape(4, 3), lin(hidden).detach().numpy(), rtol=2e-4, atol=2e-5)
