"""Network-level GPU test of the attention gate: the "Attention U-Net" graph the reference keeps commented out
in main.py:144-184, created through the `neural_net` module (fp32 engine path), one training step against
PyTorch-CPU float64 autograd on identical parameters and inputs (tests/torch_ref.py), plus the checkpoint
exchange of the gate's parameters."""
import os

import numpy as np
import pytest
import torch

import torch_ref
from util import assert_close

pytestmark = pytest.mark.gpu
os.environ.setdefault("UNET_QUIET", "1")


@pytest.fixture(scope="module")
def nn():
    import neural_net
    return neural_net


def make(nn, graph):
    net = nn.NeuralNet()
    net.set_seed(5)
    net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
    return net


def test_attention_unet_train_step_matches_torch(nn):
    graph = torch_ref.attention_unet_graph(32, 48, 3, 4)
    net = make(nn, graph)
    assert net.compute_dtype() == "f32"      # graphs with attention gates run on the fp32 kernels
    assert net.num_parameters() == torch_ref.param_count(graph)
    shapes = net.output_shapes()
    assert shapes[8] == [16, 8, 12] and shapes[12] == [8, 16, 24] and shapes[16] == [4, 32, 48]
    rng = np.random.default_rng(11)
    n = 2
    x = rng.standard_normal((n, 3, 32, 48)).astype(np.float32)
    m = (rng.random((n, 32, 48)) < 0.3).astype(np.float32)
    p0 = net.get_parameters()
    layout = {layer: (off, nwb, nbn) for layer, off, nwb, nbn in net.parameter_layout()}
    for layer in (8, 12, 16):                # gate parameters: weights He-initialised, the Ci + 1 biases zero
        off, nwb, nbn = layout[layer]
        cx, cg, ci = torch_ref.attention_dims(graph, layer)
        assert nbn == 0 and nwb == (cx + cg + 2) * ci + 1
        assert np.all(p0[off + nwb - ci - 1:off + nwb] == 0) and np.all(p0[off:off + nwb - ci - 1] != 0)
    p0 = (p0 + rng.standard_normal(p0.size).astype(np.float32) * 0.05).astype(np.float32)
    net.set_parameters(p0)

    loss_val = net.train_step(x, m, 1e-2, False)
    grads = net.get_gradients()

    params = torch_ref.split_params(graph, p0)
    acts = torch_ref.forward(graph, params, torch.from_numpy(x).double())
    ref_loss = torch_ref.dice_loss(acts[-1], torch.from_numpy(m).double())
    ref_loss.backward()
    ref_grads = torch_ref.flat_grads(graph, params)

    for i in (4, 6, 8, 12, 16, 17, 18):
        assert_close(net.layer_output(i, n), acts[i].detach().numpy(), rtol=2e-4, atol_scale=2e-5,
                     what=f"layer {i} activation")
    ref_loss = float(ref_loss.detach())
    assert abs(loss_val - ref_loss) <= 1e-4 * abs(ref_loss) + 1e-6
    for layer, (off, nwb, nbn) in layout.items():
        g, r = grads[off:off + nwb + nbn], ref_grads[off:off + nwb + nbn]
        scale = np.abs(r).max()
        assert np.abs(g - r).max() <= 1e-3 * scale + 1e-9, f"layer {layer} gradient: {np.abs(g - r).max()} vs {scale}"

    # a few steps reduce the loss; two runs give the same bits
    losses = [net.train_step(x, m, 1e-2, True) for _ in range(8)]
    assert losses[-1] < losses[0]
    net2 = make(nn, graph)
    net2.set_parameters(p0)
    net2.train_step(x, m, 1e-2, False)
    np.testing.assert_array_equal(net2.get_gradients(), grads)


def test_attention_parameters_round_trip_through_torch_layout(nn):
    from unet_amd import checkpoint
    graph = torch_ref.attention_unet_graph(16, 16, 3, 4)
    descs = torch_ref.to_layer_descs(nn, graph)
    net = nn.NeuralNet()
    net.set_seed(9)
    net.create(descs, nn.LossFunction.DICE_LOSS)
    p = net.get_parameters()
    p = (p + np.random.default_rng(0).standard_normal(p.size).astype(np.float32) * 0.1).astype(np.float32)
    net.set_parameters(p)
    state = checkpoint.to_torch(net, descs)
    cx, cg, ci = torch_ref.attention_dims(graph, 8)
    assert state["layers.8.w_x.weight"].shape == (ci, cx, 1, 1) and state["layers.8.w_g.weight"].shape == (ci, cg, 1, 1)
    assert state["layers.8.psi.weight"].shape == (1, ci, 1, 1) and state["layers.8.psi.bias"].shape == (1,)
    # the exported tensors drive a stock torch attention gate to the engine's result
    off = dict((l, o) for l, o, _, _ in net.parameter_layout())[8]
    flat = torch.from_numpy(p[off:off + (cx + cg + 2) * ci + 1]).double()
    xx = torch.randn(1, cx, 8, 8, dtype=torch.float64)
    gg = torch.randn(1, cg, 4, 4, dtype=torch.float64)
    t = {k.split(".", 2)[2]: torch.from_numpy(v).double() for k, v in state.items() if k.startswith("layers.8.")}
    q = torch.relu(torch.nn.functional.conv2d(xx, t["w_x.weight"], stride=2) +
                   torch.nn.functional.conv2d(gg, t["w_g.weight"], t["w_g.bias"]))
    alpha = torch.sigmoid(torch.nn.functional.conv2d(q, t["psi.weight"], t["psi.bias"]))[:, 0]
    # compare the coefficients through the gate's own output at the block corners (blend fraction 0)
    y = torch_ref.attention_gate(xx, gg, flat, ci)
    assert torch.allclose(y[:, :, 0, 0], xx[:, :, 0, 0] * alpha[:, 0, 0][:, None], rtol=1e-6, atol=1e-9)
    net2 = nn.NeuralNet()
    net2.set_seed(1)
    net2.create(descs, nn.LossFunction.DICE_LOSS)
    checkpoint.from_torch(net2, descs, state)
    np.testing.assert_array_equal(net2.get_parameters(), p)


def test_attention_layer_validation_messages(nn):
    L, T = nn.LayerDesc, nn.LayerType
    conv = L(T.CONV_LAYER, [8, 8, 3, 4, 3, 1, 1], [])
    with pytest.raises(RuntimeError, match="Attention layers need two parents"):
        nn.NeuralNet().create([conv, L(T.ATTENTION_LAYER, [], [0])], nn.DICE_LOSS)
    with pytest.raises(RuntimeError, match="g that is not half the size of x"):
        nn.NeuralNet().create([conv, L(T.CONV_LAYER, [8, 8, 4, 4, 3, 1, 1], [0]), L(T.ATTENTION_LAYER, [], [0, 1])],
                              nn.DICE_LOSS)


def test_attention_unet_on_the_bf16_path_matches_the_bf16_storage_oracle(nn):
    """With channel counts in multiples of 32 the same graph runs on the bf16 / NHWC engine path (the gates compute
    in fp32 between layout conversions).  Compared like the other bf16 network tests: activations against the
    float64 oracle with bf16 storage emulated (rel <= 2e-2), gradients per layer by relative L2 error and cosine."""
    from util import assert_close_bf16_net
    graph = torch_ref.attention_unet_graph(32, 48, 3, 32)
    net = make(nn, graph)
    assert net.compute_dtype() == "bf16"
    rng = np.random.default_rng(3)
    n = 4
    x = rng.standard_normal((n, 3, 32, 48)).astype(np.float32)
    m = (rng.random((n, 32, 48)) < 0.3).astype(np.float32)
    p0 = net.get_parameters()
    p0 = (p0 + rng.standard_normal(p0.size).astype(np.float32) * 0.02).astype(np.float32)
    net.set_parameters(p0)
    loss_val = net.train_step(x, m, 1e-2, False)
    grads = net.get_gradients()

    outs = {}
    for emu in (False, True):
        params = torch_ref.split_params(graph, p0)
        acts = torch_ref.forward(graph, params, torch.from_numpy(x).double(), emulate_bf16=emu)
        ref_loss = torch_ref.dice_loss(acts[-1], torch.from_numpy(m).double())
        ref_loss.backward()
        outs[emu] = ([a.detach().numpy() for a in acts], float(ref_loss.detach()), torch_ref.flat_grads(graph, params))
    acts, ref_loss, plain_grads = outs[False]
    emu_acts, emu_loss, ref_grads = outs[True]
    for i in range(len(graph)):
        assert_close_bf16_net(net.layer_output(i, n), emu_acts[i], what=f"layer {i} activation vs bf16-storage oracle")
    assert abs(loss_val - ref_loss) <= 2e-2 * abs(ref_loss) + 1e-6
    assert abs(loss_val - emu_loss) <= 2e-3 * abs(emu_loss) + 1e-6
    for layer, off, nwb, nbn in net.parameter_layout():
        co = nbn // 2
        ranges = [(off, off + nwb - co, "weights")] + ([(off + nwb, off + nwb + nbn, "gamma/beta")] if nbn else [])
        for lo, hi, what in ranges:
            g, r, pl = grads[lo:hi].astype(np.float64), ref_grads[lo:hi], plain_grads[lo:hi]
            rel = np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-12)
            cos_e = float(np.dot(g, r) / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
            cos_p = float(np.dot(g, pl) / max(np.linalg.norm(g) * np.linalg.norm(pl), 1e-30))
            # bounds from the measured mask-flip noise of this graph (tools/diag_attention_grad.py over input seeds
            # and over two summation orders of the convolution kernels: 0.08 to 0.20 per layer in every build)
            assert rel <= 0.25, f"layer {layer} {what} gradient vs bf16-storage oracle: relative L2 error {rel:.3g}"
            assert cos_e >= 0.96, f"layer {layer} {what} gradient vs bf16-storage oracle: cosine {cos_e:.4f}"
            assert cos_p >= 0.9, f"layer {layer} {what} gradient vs float64 oracle: cosine {cos_p:.4f}"
    # trains, and two runs give the same bits
    losses = [net.train_step(x, m, 1e-2, True) for _ in range(8)]
    assert losses[-1] < losses[0]
    net2 = make(nn, graph)
    net2.set_parameters(p0)
    net2.train_step(x, m, 1e-2, False)
    np.testing.assert_array_equal(net2.get_gradients(), grads)
    # inference (running statistics, BN + ReLU folded into the conv epilogues) runs through the gates too
    y = net.forward(x[:2], True)
    assert y.shape[-2:] == (32, 48) and np.isfinite(y).all()


def test_attention_unet_checkpoint_round_trip(nn, tmp_path):
    """save() / load() of a network with attention gates (layers with weights but no batch norm): a restored
    network continues training with the same bits."""
    from unet_amd import checkpoint
    graph = torch_ref.attention_unet_graph(16, 16, 3, 32)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 3, 16, 16)).astype(np.float32)
    m = (rng.random((2, 16, 16)) < 0.3).astype(np.float32)
    net = make(nn, graph)
    for _ in range(3):
        net.train_step(x, m, 1e-2, True)
    path = str(tmp_path / "attn.npz")
    checkpoint.save(net, path)
    net2 = make(nn, graph)
    checkpoint.load(net2, path)
    np.testing.assert_array_equal(net2.get_parameters(), net.get_parameters())
    a, b = net.train_step(x, m, 1e-2, True), net2.train_step(x, m, 1e-2, True)
    assert a == b
    np.testing.assert_array_equal(net2.get_parameters(), net.get_parameters())
