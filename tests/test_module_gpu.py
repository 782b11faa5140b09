"""GPU tests of the Python module `neural_net` (the drop-in boundary):
API behaviour the reference's scripts rely on (main.py, test.py,
test/test_convolution.py) and network-level parity against the PyTorch-CPU
oracle (float64) for forward, loss, every gradient, running statistics and the
Adam update.  fp32 path: relative <= 1e-4 on op outputs; gradients that went
through many fp32 layers are held to 1e-3 of the gradient's magnitude.
"""
import os

import numpy as np
import pytest
import torch

import torch_ref
from util import assert_close, assert_exact

pytestmark = pytest.mark.gpu
os.environ.setdefault("UNET_QUIET", "1")


@pytest.fixture(scope="module")
def nn():
    import neural_net
    return neural_net


def _make(nn, graph, loss=None, seed=3):
    net = nn.NeuralNet()
    net.set_seed(seed)
    net.create(torch_ref.to_layer_descs(nn, graph), loss if loss is not None else nn.LossFunction.DICE_LOSS)
    return net


# ------------------------------------------------------------ Convolution ----
def test_convolution2d_config1(nn, golden):
    """BASELINE config 1: 1x1x64x64 image, edge kernel, pad 1, stride 1 vs PyTorch-CPU."""
    conv = nn.Convolution()
    x = golden["conv_g_x"][0]                       # [1,64,64]
    kernel = np.array([[-1, 0, 1]] * 3).reshape(1, 1, 3, 3)   # int64 on purpose (test_convolution.py:139-142)
    out = conv.convolution2D(x, kernel, np.array([0]), 1, 1)
    assert out.dtype == np.float32 and out.shape == (64 * 64,)
    assert_close(out.reshape(1, 64, 64), golden["conv_g_y"][0], what="convolution2D")


def test_convolution2d_multi_channel_and_errors(nn, golden):
    conv = nn.Convolution()
    n, ci, co, h, w, k, p, s = (int(v) for v in golden["conv_c_cfg"])
    wt = golden["conv_c_w"]
    out = conv.convolution2D(golden["conv_c_x"][0], wt[:ci * co * k * k].reshape(ci, co, k, k), wt[-co:], p, s)
    assert_close(out.reshape(golden["conv_c_y"][0].shape), golden["conv_c_y"][0], what="convolution2D stride 2")
    with pytest.raises(RuntimeError, match="Only accepting single images"):
        conv.convolution2D(np.zeros((1, 1, 4, 4), np.float32), np.zeros((1, 1, 3, 3)), np.zeros(1), 1, 1)
    with pytest.raises(RuntimeError, match=r"\[C_in\]\[C_out\]\[H\]\[W\]"):
        conv.convolution2D(np.zeros((1, 4, 4), np.float32), np.zeros((3, 3)), np.zeros(1), 1, 1)
    with pytest.raises(RuntimeError, match="Kernel C_in must match input C_in"):
        conv.convolution2D(np.zeros((2, 4, 4), np.float32), np.zeros((1, 1, 3, 3)), np.zeros(1), 1, 1)
    with pytest.raises(RuntimeError, match="square and odd"):
        conv.convolution2D(np.zeros((1, 4, 4), np.float32), np.zeros((1, 1, 2, 2)), np.zeros(1), 1, 1)


def test_convolution_train_matches_reference_recipe(nn, oracle):
    """Convolution.train = 100 steps of conv -> difference loss -> conv backward
    -> Adam(t) (convolution_numpy.h:84-135), checked against the same recipe
    evaluated with the CPU oracle."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 24, 24)).astype(np.float32)
    edge = np.array([[-1, 0, 1]] * 3, np.float32).reshape(1, 1, 3, 3)
    target = oracle.convolve(x[None], np.concatenate([edge.ravel(), [0]]).astype(np.float32), 1, 3, 1, 1)[0]
    k0 = np.array([[0.0749, 0.4010, 0.0577], [-0.1607, 0.1313, -0.1326], [0.2074, 0.4494, 0.2920]], np.float32)
    b0 = np.array([0.1970], np.float32)
    lr = 0.35
    got = nn.Convolution().train(x, target, k0.reshape(1, 1, 3, 3), b0, 1, 1, lr)
    w = np.concatenate([k0.ravel(), b0]).astype(np.float32)
    st = np.zeros(2 * w.size, np.float32)
    for t in range(1, 101):
        y = oracle.convolve(x[None], w, 1, 3, 1, 1)
        g = oracle.difference_backward(target[None], y)
        _, dw = oracle.convolve_backward(x[None], g, w, 3, 1, 1)
        w, st = oracle.adam(w, dw, st, lr, t)
    assert got.shape == (10,)
    assert_close(got, w, rtol=2e-3, atol_scale=1e-4, what="Convolution.train weights after 100 Adam steps")
    # the fit must have moved towards the edge kernel
    assert np.abs(got[:9] - edge.ravel()).mean() < np.abs(k0.ravel() - edge.ravel()).mean()


# ---------------------------------------------------------------- UNetTest ----
def test_unet_selftests_pass(nn, capfd):
    t = nn.UNetTest()
    t.testMaxPool()
    t.testBatchNorm()
    t.testConvolution()
    t.testUpsample()
    out = capfd.readouterr().out
    for name in ["Max pool tests pass!", "Batch Norm tests pass!", "Convolution tests pass!", "Upsample tests pass!"]:
        assert name in out


# --------------------------------------------------------------- NeuralNet ----
def test_create_validation_messages(nn):
    L, T = nn.LayerDesc, nn.LayerType
    with pytest.raises(RuntimeError, match="invalid parent index"):
        nn.NeuralNet().create([L(T.CONV_LAYER, [8, 8, 3, 4, 3, 1, 1], [0])], nn.DICE_LOSS)
    with pytest.raises(RuntimeError, match="single output node"):
        nn.NeuralNet().create([L(T.CONV_LAYER, [8, 8, 3, 4, 3, 1, 1], []),
                               L(T.CONV_LAYER, [8, 8, 3, 4, 3, 1, 1], [])], nn.DICE_LOSS)
    with pytest.raises(RuntimeError, match="Concat layer must have two parents"):
        nn.NeuralNet().create([L(T.CONV_LAYER, [8, 8, 3, 4, 3, 1, 1], []), L(T.CONCAT_LAYER, [], [0])], nn.DICE_LOSS)
    # a smaller first parent, or odd margins, cannot be centre-cropped to the second parent's size
    with pytest.raises(RuntimeError, match="equal heights and widths"):
        nn.NeuralNet().create([L(T.CONV_LAYER, [8, 8, 3, 4, 3, 1, 1], []), L(T.MAX_POOL_LAYER, [4, 8, 8, 2, 2], [0]),
                               L(T.CONCAT_LAYER, [], [1, 0])], nn.DICE_LOSS)
    with pytest.raises(RuntimeError, match="equal heights and widths"):
        nn.NeuralNet().create([L(T.CONV_LAYER, [9, 9, 3, 4, 3, 1, 1], []), L(T.CONV_LAYER, [9, 9, 4, 4, 3, 0, 2], [0]),
                               L(T.CONCAT_LAYER, [], [0, 1])], nn.DICE_LOSS)
    with pytest.raises(RuntimeError, match="does not equal output shape of parent"):
        nn.NeuralNet().create([L(T.CONV_LAYER, [8, 8, 3, 4, 3, 1, 1], []), L(T.UPSAMPLING_LAYER, [8, 8, 5, 4, 2], [0])],
                              nn.DICE_LOSS)
    with pytest.raises(RuntimeError, match="cannot have more than one parent"):
        nn.NeuralNet().create([L(T.CONV_LAYER, [8, 8, 3, 4, 3, 1, 1], []), L(T.CONV_LAYER, [8, 8, 4, 4, 3, 1, 1], [0]),
                               L(T.CONV_LAYER, [8, 8, 4, 4, 3, 1, 1], [0, 1])], nn.DICE_LOSS)
    with pytest.raises(RuntimeError, match="FC layers need 4 shape values"):
        nn.NeuralNet().create([L(T.CONV_LAYER, [8, 8, 3, 4, 3, 1, 1], []), L(T.FC_LAYER, [], [0])], nn.DICE_LOSS)
    with pytest.raises(RuntimeError, match="does not equal output shape of parent"):
        nn.NeuralNet().create([L(T.CONV_LAYER, [8, 8, 3, 4, 3, 1, 1], []), L(T.FC_LAYER, [4, 8, 7, 2], [0])], nn.DICE_LOSS)
    with pytest.raises(RuntimeError, match="BN layers need exactly one parent"):
        nn.NeuralNet().create([L(T.BN_LAYER, [], [])], nn.DICE_LOSS)
    with pytest.raises(RuntimeError, match="does not equal output shape of parent"):
        nn.NeuralNet().create([L(T.CONV_LAYER, [8, 8, 3, 4, 3, 1, 1], []), L(T.BN_LAYER, [5, 8, 8], [0])], nn.DICE_LOSS)
    net = nn.NeuralNet()
    with pytest.raises(RuntimeError, match="create"):
        net.train(np.zeros((1, 3, 8, 8), np.float32), np.zeros((1, 8, 8), np.float32), 1e-3, 1, 1)


def test_shapes_and_parameter_layout(nn):
    graph = torch_ref.main_py_graph(32, 32, 3, 4)
    net = _make(nn, graph)
    shapes = net.output_shapes()
    assert shapes[0] == [4, 32, 32] and shapes[1] == [4, 16, 16] and shapes[6] == [32, 4, 4]
    assert shapes[7] == [16, 8, 8] and shapes[8] == [32, 8, 8] and shapes[14] == [8, 32, 32] and shapes[15] == [1, 32, 32]
    assert net.num_parameters() == torch_ref.param_count(graph)
    p = net.get_parameters()
    for layer, off, nwb, nbn in net.parameter_layout():
        co = nbn // 2
        assert np.all(p[off + nwb - co:off + nwb] == 0), "biases start at zero"
        assert_exact(p[off + nwb:off + nwb + nbn], np.tile([1, 0], co).astype(np.float32), what="gamma=1, beta=0")
        w = p[off:off + nwb - co]
        kind, s = graph[layer][0], graph[layer][1]
        fan = s[2] * (9 if kind == "conv" else 1)
        assert abs(w.std() / np.sqrt(2.0 / fan) - 1) < 0.25, "He(fan-in) scale"
        assert_exact(net.running_stats(layer), np.tile([0, 1], co).astype(np.float32), what="running mean 0, var 1")


@pytest.mark.parametrize("loss_name", ["dice", "difference"])
def test_train_step_matches_torch(nn, loss_name):
    """One full fwd+bwd+Adam step of the main.py graph (scaled down) against
    PyTorch-CPU float64 autograd on identical parameters and inputs."""
    graph = torch_ref.main_py_graph(16, 24, 3, 4)
    loss = nn.LossFunction.DICE_LOSS if loss_name == "dice" else nn.LossFunction.DIFFERENCE_LOSS
    net = _make(nn, graph, loss)
    rng = np.random.default_rng(7)
    n = 3
    x = rng.standard_normal((n, 3, 16, 24)).astype(np.float32)
    m = (rng.random((n, 16, 24)) < 0.3).astype(np.float32)
    p0 = net.get_parameters()
    # perturb gamma/beta/bias so that they matter
    p0 = (p0 + rng.standard_normal(p0.size).astype(np.float32) * 0.05).astype(np.float32)
    net.set_parameters(p0)

    loss_val = net.train_step(x, m, 1e-2, False)
    grads = net.get_gradients()

    params = torch_ref.split_params(graph, p0)
    acts = torch_ref.forward(graph, params, torch.from_numpy(x).double())
    ref_loss = (torch_ref.dice_loss if loss_name == "dice" else torch_ref.difference_loss)(
        acts[-1], torch.from_numpy(m).double())
    ref_loss.backward()
    ref_grads = torch_ref.flat_grads(graph, params)

    for i in (0, 1, 6, 8, 15):
        assert_close(net.layer_output(i, n), acts[i].detach().numpy(), rtol=2e-4, atol_scale=2e-5,
                     what=f"layer {i} activation")
    assert abs(loss_val - float(ref_loss)) <= 1e-4 * abs(float(ref_loss)) + 1e-6
    # gradients, layer by layer
    for layer, off, nwb, nbn in net.parameter_layout():
        g, r = grads[off:off + nwb + nbn], ref_grads[off:off + nwb + nbn]
        scale = np.abs(r).max()
        assert np.abs(g - r).max() <= 1e-3 * scale + 1e-9, f"layer {layer} gradient: {np.abs(g - r).max()} vs {scale}"

    # Adam step on the same gradients
    net2_loss = net.train_step(x, m, 1e-2, True)
    assert abs(net2_loss - loss_val) <= 1e-6 + 1e-5 * abs(loss_val)
    p1 = net.get_parameters()
    state = (np.zeros(p0.size), np.zeros(p0.size))
    ref_p1 = torch_ref.adam_step(p0.astype(np.float64), ref_grads.astype(np.float64), state, 1e-2, 1)
    # Adam's first step moves every weight by ~lr*sign(g): compare where the gradient is not ~0
    big = np.abs(ref_grads) > 1e-6 * np.abs(ref_grads).max()
    assert_close(p1[big], ref_p1[big], rtol=1e-3, atol_scale=1e-4, what="parameters after one Adam step")


def test_running_stats_and_inference(nn):
    graph = torch_ref.main_py_graph(16, 16, 3, 4)
    net = _make(nn, graph)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((4, 3, 16, 16)).astype(np.float32)
    m = (rng.random((4, 16, 16)) < 0.3).astype(np.float32)
    net.train_step(x, m, 1e-3, False)
    params = torch_ref.split_params(graph, net.get_parameters())
    z = torch.nn.functional.conv2d(torch.from_numpy(x).double(), params[0]["w"].permute(1, 0, 2, 3), params[0]["b"],
                                   padding=1)
    mean, var = z.mean(dim=(0, 2, 3)), z.var(dim=(0, 2, 3), unbiased=False)
    want = torch.stack([0.1 * mean, 0.9 + 0.1 * var], dim=1).reshape(-1).detach().numpy()
    assert_close(net.running_stats(0), want, what="running stats after one step")
    # inference forward uses the running statistics
    running = [net.running_stats(i) for i in range(len(graph))]
    y_inf = net.forward(x[:2], True)
    acts = torch_ref.forward(graph, params, torch.from_numpy(x[:2]).double(), training=False, running=running)
    assert_close(y_inf, acts[-1].detach().numpy(), rtol=2e-4, atol_scale=2e-5, what="inference forward")
    out = net.infer(x[0])
    assert out.shape == (1, 16, 16) and out.dtype == np.float32
    assert_exact(out, (net.forward(x[:1], True)[0] >= 0.5).astype(np.float32), what="infer = thresholded forward")
    d = net.test(x[0], (m[0] > 0).astype(np.uint8))
    pred = out[0]
    want_d = 2 * (pred * m[0]).sum() / (pred.sum() + m[0].sum() + 1e-5)
    assert abs(d - want_d) < 1e-5
    with pytest.raises(RuntimeError, match="Only accepting single images"):
        net.infer(x)
    with pytest.raises(RuntimeError, match="Mask dimensions do not match"):
        net.test(x[0], np.zeros((8, 8), np.uint8))


def test_train_loop_learns_and_handles_tail_batch(nn):
    """main.py's call: train(images, masks, lr, epochs, batch) with N not a
    multiple of the batch size; the loss must go down on a learnable toy task."""
    graph = torch_ref.main_py_graph(16, 16, 3, 4)
    net = _make(nn, graph)
    rng = np.random.default_rng(2)
    n = 10
    x = rng.standard_normal((n, 3, 16, 16)).astype(np.float32)
    m = (x[:, 0] > 0.3).astype(np.float32)       # mask is a simple function of the image
    net.train(x, m, 1e-2, 1, 4)
    first = net.last_epoch_loss()
    net.train(x, m, 1e-2, 12, 4)
    last = net.last_epoch_loss()
    assert np.isfinite(first) and np.isfinite(last)
    assert last < first - 0.05, (first, last)
    with pytest.raises(RuntimeError, match="Mask dimensions do not equal image dimensions"):
        net.train(x, m[:, :8], 1e-2, 1, 4)


def test_deterministic(nn):
    graph = torch_ref.main_py_graph(16, 16, 3, 4)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((4, 3, 16, 16)).astype(np.float32)
    m = (rng.random((4, 16, 16)) < 0.3).astype(np.float32)
    outs = []
    for _ in range(2):
        net = _make(nn, graph, seed=11)
        net.train_step(x, m, 1e-2, True)
        net.train_step(x, m, 1e-2, True)
        outs.append((net.get_parameters(), net.get_gradients()))
    assert_exact(outs[0][0], outs[1][0], what="parameters bitwise reproducible")
    assert_exact(outs[0][1], outs[1][1], what="gradients bitwise reproducible")
