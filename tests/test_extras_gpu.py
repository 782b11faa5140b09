"""LEAKY_ReLU / TANH activations and the cropping concat: the HIP entry points against the CPU oracle, and a
small U-Net of unpadded convolutions (the geometry of BASELINE config 4's "original U-Net": valid 3x3
convolutions, skip tensors centre-cropped at the concat, output smaller than the input) with those
activations against the PyTorch float64 oracle (fp32 bar: 1e-4)."""
import os

import numpy as np
import pytest
import torch

import torch_ref
from gpu_util import dev, empty, host, p
from unet_amd import capi
from util import assert_close

pytestmark = pytest.mark.gpu
os.environ.setdefault("UNET_QUIET", "1")


def test_leaky_relu_and_tanh_ops(oracle):
    rng = np.random.default_rng(2)
    xh = rng.standard_normal((2, 6, 9, 11)).astype(np.float32)
    dyh = rng.standard_normal(xh.shape).astype(np.float32)
    x, dy = dev(xh), dev(dyh)
    n = xh.size
    for slope in (0.01, 0.3):
        y, dx = empty(xh.shape), empty(xh.shape)
        capi.unet_leaky_relu_f32(p(x), p(y), n, slope, None)
        capi.unet_leaky_relu_backward_f32(p(dy), p(x), p(dx), n, slope, None)
        np.testing.assert_array_equal(host(y), oracle.leaky_relu(xh, slope))
        np.testing.assert_array_equal(host(dx), oracle.leaky_relu_backward(dyh, xh, slope))
    y, dx = empty(xh.shape), empty(xh.shape)
    capi.unet_tanh_f32(p(x), p(y), n, None)
    capi.unet_tanh_backward_f32(p(dy), p(y), p(dx), n, None)
    assert_close(host(y), oracle.tanh(xh), what="tanh")
    assert_close(host(dx), oracle.tanh_backward(dyh, host(y)), what="tanh backward")


@pytest.mark.parametrize("sizes", [(12, 10, 8, 6), (9, 9, 9, 9), (37, 41, 31, 33)])
def test_concat_crop_ops(oracle, sizes):
    ha, wa, hb, wb = sizes
    rng = np.random.default_rng(3)
    n, c1, c2 = 3, 5, 7
    ah = rng.standard_normal((n, c1, ha, wa)).astype(np.float32)
    bh = rng.standard_normal((n, c2, hb, wb)).astype(np.float32)
    out = empty((n, c1 + c2, hb, wb))
    a, b = dev(ah), dev(bh)
    capi.unet_concat_crop_f32(p(a), p(b), p(out), n, c1, c2, ha, wa, hb, wb, None)
    np.testing.assert_array_equal(host(out), oracle.concat_crop(ah, bh))          # layout: bit-exact
    dyh = rng.standard_normal((n, c1 + c2, hb, wb)).astype(np.float32)
    da, db = empty(ah.shape), empty(bh.shape)
    dy = dev(dyh)
    capi.unet_concat_crop_backward_f32(p(dy), p(da), p(db), n, c1, c2, ha, wa, hb, wb, None)
    da_o, db_o = oracle.concat_crop_backward(dyh, c1, ha, wa)
    np.testing.assert_array_equal(host(da), da_o)
    np.testing.assert_array_equal(host(db), db_o)


@pytest.mark.parametrize("acts", [("relu", "relu"), ("leaky:0.1", "tanh")])
def test_unpadded_unet_with_cropped_skips_matches_torch(acts):
    import neural_net as nn
    graph, ho, wo = torch_ref.valid_unet_graph(60, 60, 3, 8, 2, acts)
    assert (ho, wo) == (20, 20)
    net = nn.NeuralNet()
    net.set_seed(5)
    net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
    assert net.compute_dtype() == "f32"          # unpadded convolutions run on the fp32 kernels
    p0 = net.get_parameters()
    # A ReLU / leaky-ReLU argument within fp32 rounding of zero gets its mask decided differently by the fp32
    # engine and the float64 oracle; at this size ONE such element moves the gradient by 4e-3 of its norm.
    # The comparison below is made on an input whose smallest such argument is far from that.
    for seed in range(4, 40):
        rng = np.random.default_rng(seed)
        x = rng.standard_normal((3, 3, 60, 60)).astype(np.float32)
        m = (rng.random((3, ho, wo)) < 0.3).astype(np.float32)
        params = torch_ref.split_params(graph, p0)
        pre = []
        acts_t = torch_ref.forward(graph, params, torch.from_numpy(x).double(), training=True, pre=pre)
        if min(t.abs().min().item() for t in pre[:-1]) > 5e-6:
            break
    else:
        raise AssertionError("no seed keeps the activations' arguments away from zero")
    with pytest.raises(RuntimeError, match="Mask dimensions"):
        net.train_step(x, np.zeros((3, 60, 60), np.float32), 1e-3, False)
    loss = net.train_step(x, m, 1e-3, False)
    g = net.get_gradients().astype(np.float64)
    assert tuple(acts_t[-1].shape) == (3, 1, ho, wo)
    ref_loss = torch_ref.dice_loss(acts_t[-1], torch.from_numpy(m).double())
    ref_loss.backward()
    ref_g = torch_ref.flat_grads(graph, params)
    assert abs(loss - ref_loss.item()) <= 1e-4 * abs(ref_loss.item()) + 1e-6
    assert_close(net.layer_output(len(graph) - 1, 3), acts_t[-1].detach().numpy(), what="output")
    assert np.linalg.norm(g - ref_g) <= 1e-4 * np.linalg.norm(ref_g), np.linalg.norm(g - ref_g) / np.linalg.norm(ref_g)
    # and it trains: a few steps lower the loss
    for _ in range(10):
        last = net.train_step(x, m, 5e-3, True)
    assert last < loss


def test_config4_literal_geometry_is_accepted():
    """BASELINE config 4 as written: 572 x 572 tiles through four levels of unpadded double convolutions give
    the 388 x 388 segmentation map of the original U-Net; one forward + backward runs and is finite."""
    import neural_net as nn
    graph, ho, wo = torch_ref.valid_unet_graph(572, 572, 3, 64, 4)
    assert (ho, wo) == (388, 388)
    net = nn.NeuralNet()
    net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
    rng = np.random.default_rng(6)
    x = rng.standard_normal((1, 3, 572, 572)).astype(np.float32)
    m = (rng.random((1, ho, wo)) < 0.1).astype(np.float32)
    loss = net.train_step(x, m, 1e-3, True)
    assert np.isfinite(loss) and np.isfinite(net.get_gradients()).all()
    assert net.infer(x[0]).shape == (1, 388, 388)


@pytest.mark.parametrize("loss_name", ["difference", "dice"])
def test_epoch_metric_is_the_loss_being_minimised(loss_name):
    """train() reports the mean over the epoch's batches of the loss it minimises, also for the difference
    loss and with a short last batch; the first epoch on a fresh network (its staging buffers are allocated
    while the first batch is already running) is the case checked."""
    import neural_net as nn
    rng = np.random.default_rng(9)
    graph = torch_ref.main_py_graph(16, 24, 3, 4)
    net = nn.NeuralNet()
    net.set_seed(2)
    net.create(torch_ref.to_layer_descs(nn, graph),
               nn.LossFunction.DICE_LOSS if loss_name == "dice" else nn.LossFunction.DIFFERENCE_LOSS)
    x = rng.standard_normal((6, 3, 16, 24)).astype(np.float32)
    mk = (rng.random((6, 16, 24)) < 0.3).astype(np.float32)
    net.train(x, mk, 0.0, 1, 4)                             # learning rate 0: the parameters stay
    got = net.last_epoch_loss()
    per_batch = [net.train_step(x[i:i + 4], mk[i:i + 4], 0.0, False) for i in (0, 4)]
    want = (4 * per_batch[0] + 2 * per_batch[1]) / 6 if loss_name == "dice" else np.mean(per_batch)
    assert abs(got - want) <= 1e-5 * want, (got, want, per_batch)


def test_difference_loss_value():
    """unet_difference_loss_sums_f32 against numpy, and the loss train_step() returns under DIFFERENCE_LOSS
    (0.5 * mean (y - m)^2) against the PyTorch float64 oracle."""
    import neural_net as nn
    rng = np.random.default_rng(8)
    n, h, w = 5, 70, 90
    mh = (rng.random((n, h, w)) < 0.4).astype(np.float32)
    ph = rng.random((n, h, w)).astype(np.float32)
    m, pr, lv = dev(mh), dev(ph), empty((n, 3))
    capi.unet_difference_loss_sums_f32(p(m), p(pr), h, w, n, p(lv), None)
    want = np.stack([((ph - mh).astype(np.float64) ** 2).sum(axis=(1, 2)), mh.sum(axis=(1, 2)), ph.sum(axis=(1, 2))], 1)
    assert_close(host(lv), want, what="difference loss sums")

    graph = torch_ref.main_py_graph(16, 24, 3, 4)
    net = nn.NeuralNet()
    net.set_seed(2)
    net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DIFFERENCE_LOSS)
    x = rng.standard_normal((4, 3, 16, 24)).astype(np.float32)
    mk = (rng.random((4, 16, 24)) < 0.3).astype(np.float32)
    got = net.train_step(x, mk, 0.0, False)
    acts = torch_ref.forward(graph, torch_ref.split_params(graph, net.get_parameters()),
                             torch.from_numpy(x).double(), training=True)
    ref = torch_ref.difference_loss(acts[-1], torch.from_numpy(mk).double()).item()
    assert abs(got - ref) <= 1e-4 * ref


def test_nonfinite_gradient_watch(oracle):
    """unet_flag_nonfinite_f32 records the first call that saw a NaN or Inf; the module raises from train_step()
    when the watch is on (set_check_nan / UNET_CHECK_NAN=1) and says at which optimizer step."""
    import neural_net as nn
    xh = np.linspace(-1, 1, 1003).astype(np.float32)
    x = dev(xh)
    flag = torch.zeros(1, dtype=torch.int32, device=x.device)
    capi.unet_flag_nonfinite_f32(p(x), xh.size, p(flag), 5, None)
    assert int(flag.item()) == 0
    for pos, bad in ((1002, np.inf), (17, np.nan), (400, -np.inf)):
        y = xh.copy()
        y[pos] = bad
        flag.zero_()
        yd = dev(y)
        capi.unet_flag_nonfinite_f32(p(yd), y.size, p(flag), 7, None)
        capi.unet_flag_nonfinite_f32(p(yd), y.size, p(flag), 9, None)      # the first record stays
        assert int(flag.item()) == 7

    graph = torch_ref.main_py_graph(16, 16, 3, 4)
    net = nn.NeuralNet()
    net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
    net.set_check_nan(True)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 3, 16, 16)).astype(np.float32)
    m = (rng.random((2, 16, 16)) < 0.3).astype(np.float32)
    net.train_step(x, m, 1e-3, True)
    net.train_step(x, m, 1e-3, True)
    assert net.nonfinite_step() == 0
    x[1, 2, 3, 4] = np.nan
    with pytest.raises(RuntimeError, match="first seen at optimizer step 3"):
        net.train_step(x, m, 1e-3, True)
    assert net.nonfinite_step() == 3


def test_cross_entropy_loss_in_the_engine():
    """LossFunction.CROSS_ENTROPY_LOSS (not in the reference, whose SOFTMAX is an enum value only): a final layer
    with several channels and the SOFTMAX activation gives logits, masks hold class indices; loss, gradients and a
    few Adam steps against the PyTorch float64 oracle (F.cross_entropy), class map from infer()."""
    import neural_net as nn
    import torch.nn.functional as F
    classes = 4
    graph = torch_ref.main_py_graph(16, 24, 3, 4)
    kind, shape, parents, _ = graph[-1]
    graph[-1] = (kind, shape[:3] + [classes] + shape[4:], parents, "logits")
    descs = torch_ref.to_layer_descs(nn, graph[:-1])
    descs.append(nn.LayerDesc(nn.LayerType.CONV_LAYER, graph[-1][1], parents, nn.ActivationType.SOFTMAX))
    net = nn.NeuralNet()
    net.set_seed(4)
    net.create(descs, nn.LossFunction.CROSS_ENTROPY_LOSS)
    assert net.compute_dtype() == "f32"
    rng = np.random.default_rng(10)
    x = rng.standard_normal((3, 3, 16, 24)).astype(np.float32)
    labels = rng.integers(0, classes, (3, 16, 24)).astype(np.float32)
    p0 = net.get_parameters()
    loss = net.train_step(x, labels, 1e-2, False)
    g = net.get_gradients().astype(np.float64)
    params = torch_ref.split_params(graph, p0)
    acts = torch_ref.forward(graph, params, torch.from_numpy(x).double(), training=True)
    ref = F.cross_entropy(acts[-1], torch.from_numpy(labels).long())
    ref.backward()
    ref_g = torch_ref.flat_grads(graph, params)
    assert abs(loss - ref.item()) <= 1e-4 * ref.item()
    assert np.linalg.norm(g - ref_g) <= 1e-3 * np.linalg.norm(ref_g), np.linalg.norm(g - ref_g) / np.linalg.norm(ref_g)
    for _ in range(30):
        last = net.train_step(x, labels, 2e-2, True)
    assert last < 0.7 * loss
    net.train(x, labels, 1e-3, 1, 2)
    assert 0 < net.last_epoch_loss() < loss
    cls = net.infer(x[0])
    assert cls.shape == (1, 16, 24) and set(np.unique(cls)).issubset(set(range(classes)))


@pytest.mark.parametrize("classes", [4, 32])
def test_cross_entropy_loss_on_the_bf16_path(classes):
    """The same loss with every hidden channel count a multiple of 32: the graph runs on the bf16 kernels, the
    final layer's logits land as fp32 [N][C][H][W] (the narrow route) and the softmax cross-entropy, its gradient
    and the class map are the fp32 kernels' -- against the PyTorch float64 oracle that rounds to bf16 where the
    engine stores bf16 (loss 2e-3, output and parameter gradients 2e-2), then a few Adam steps."""
    import neural_net as nn
    import torch.nn.functional as F
    graph = torch_ref.main_py_graph(32, 32, 3, 32)
    kind, shape, parents, _ = graph[-1]
    graph[-1] = (kind, shape[:3] + [classes] + shape[4:], parents, "logits")
    descs = torch_ref.to_layer_descs(nn, graph[:-1])
    descs.append(nn.LayerDesc(nn.LayerType.CONV_LAYER, graph[-1][1], parents, nn.ActivationType.SOFTMAX))
    net = nn.NeuralNet()
    net.set_seed(5)
    net.create(descs, nn.LossFunction.CROSS_ENTROPY_LOSS)
    assert net.compute_dtype() == "bf16"
    rng = np.random.default_rng(11)
    x = rng.standard_normal((4, 3, 32, 32)).astype(np.float32)
    labels = rng.integers(0, classes, (4, 32, 32)).astype(np.float32)
    labels[0, :4, :4] = 255          # an ignore label: no loss, no gradient, left out of the mean
    p0 = net.get_parameters()
    p0 = (p0 + rng.standard_normal(p0.size).astype(np.float32) * 0.02).astype(np.float32)
    net.set_parameters(p0)
    loss = net.train_step(x, labels, 1e-2, False)
    g = net.get_gradients().astype(np.float64)
    logits = net.layer_output(len(graph) - 1, 4)
    params = torch_ref.split_params(graph, p0)
    acts = torch_ref.forward(graph, params, torch.from_numpy(x).double(), training=True, emulate_bf16=True)
    target = torch.from_numpy(np.where(labels >= classes, -100, labels)).long()
    ref = F.cross_entropy(acts[-1], target, ignore_index=-100)
    ref.backward()
    ref_g = torch_ref.flat_grads(graph, params)
    ref_logits = acts[-1].detach().numpy()
    assert abs(loss - ref.item()) <= 2e-3 * max(1.0, ref.item()), (loss, ref.item())
    assert np.linalg.norm(logits - ref_logits) <= 2e-2 * np.linalg.norm(ref_logits)
    rel = np.linalg.norm(g - ref_g) / np.linalg.norm(ref_g)
    cos = float(g @ ref_g / (np.linalg.norm(g) * np.linalg.norm(ref_g)))
    assert rel <= 0.15 and cos >= 0.99, (rel, cos)     # the deep-graph bars of tests/test_module_bf16_gpu.py
    for _ in range(30):
        last = net.train_step(x, labels, 2e-2, True)
    assert last < 0.8 * loss, (last, loss)
    cls = net.infer(x[0])
    assert cls.shape == (1, 32, 32) and set(np.unique(cls)).issubset(set(range(classes)))


def _torch_bn(x, gb, eps=1e-8):
    import torch.nn.functional as F
    return F.batch_norm(x, None, None, gb[0::2], gb[1::2], training=True, eps=eps)


def test_bn_layer_and_fc_layer_against_torch():
    """BN_LAYER and FC_LAYER, which the reference declares and leaves empty (layer.h:58-66): a batch norm of its own
    with the descriptor's activation, and a fully connected layer over the whole [C,H,W] input -- here as the
    classification head of a small network under the cross-entropy loss (labels [N,1,1]).  Loss and every parameter
    gradient against PyTorch float64; then the loss falls under Adam."""
    import neural_net as nn
    import torch.nn.functional as F
    L, T, A = nn.LayerDesc, nn.LayerType, nn.ActivationType
    classes, n = 5, 6
    # (a sigmoid on the first layer: a ReLU's exact zeros would tie in the pooling windows, where the engine follows
    # the reference -- every tied maximum receives the gradient -- and PyTorch picks one)
    descs = [L(T.CONV_LAYER, [8, 8, 3, 4, 3, 1, 1], [], A.SIGMOID),
             L(T.BN_LAYER, [], [0], A.TANH),
             L(T.MAX_POOL_LAYER, [4, 8, 8, 2, 2], [1]),
             L(T.BN_LAYER, [4, 4, 4], [2]),                       # NONE: no activation
             L(T.FC_LAYER, [4, 4, 4, 7], [3], A.LEAKY_ReLU, 0.1),
             L(T.FC_LAYER, [7, 1, 1, classes], [4], A.SOFTMAX)]
    net = nn.NeuralNet()
    net.set_seed(2)
    net.create(descs, nn.LossFunction.CROSS_ENTROPY_LOSS)
    assert net.compute_dtype() == "f32"
    assert net.output_shapes() == [[4, 8, 8], [4, 8, 8], [4, 4, 4], [4, 4, 4], [7, 1, 1], [classes, 1, 1]]
    layout = net.parameter_layout()
    assert [(i, wb, bn) for i, _, wb, bn in layout] == [(0, 3 * 4 * 9 + 4, 8), (1, 0, 8), (3, 0, 8), (4, 64 * 7 + 7, 0),
                                                         (5, 7 * classes + classes, 0)]
    rng = np.random.default_rng(21)
    x = rng.standard_normal((n, 3, 8, 8)).astype(np.float32)
    labels = rng.integers(0, classes, (n, 1, 1)).astype(np.float32)
    p0 = net.get_parameters()
    p0 = (p0 + rng.standard_normal(p0.size).astype(np.float32) * 0.1).astype(np.float32)
    net.set_parameters(p0)
    loss = net.train_step(x, labels, 1e-2, False)
    g = net.get_gradients().astype(np.float64)

    flat = torch.from_numpy(p0.astype(np.float64)).requires_grad_(True)
    off = {i: o for i, o, _, _ in layout}
    w0 = flat[off[0]:off[0] + 108].reshape(3, 4, 3, 3)
    b0 = flat[off[0] + 108:off[0] + 112]
    t = F.conv2d(torch.from_numpy(x).double(), w0.permute(1, 0, 2, 3), b0, padding=1)
    t = torch.sigmoid(_torch_bn(t, flat[off[0] + 112:off[0] + 120]))
    t = torch.tanh(_torch_bn(t, flat[off[1]:off[1] + 8]))
    t = F.max_pool2d(t, 2, 2)
    t = _torch_bn(t, flat[off[3]:off[3] + 8])
    t = t.reshape(n, 64) @ flat[off[4]:off[4] + 448].reshape(64, 7) + flat[off[4] + 448:off[4] + 455]
    t = F.leaky_relu(t, 0.1)
    logits = t @ flat[off[5]:off[5] + 7 * classes].reshape(7, classes) + flat[off[5] + 7 * classes:off[5] + 8 * classes]
    ref = F.cross_entropy(logits, torch.from_numpy(labels.reshape(n)).long())
    ref.backward()
    ref_g = flat.grad.numpy()
    assert abs(loss - ref.item()) <= 1e-4 * ref.item(), (loss, ref.item())
    per_layer = {i: float(np.linalg.norm(g[o:o + wb + bn] - ref_g[o:o + wb + bn]) / np.linalg.norm(ref_g[o:o + wb + bn]))
                 for i, o, wb, bn in layout}
    assert max(per_layer.values()) <= 1e-3, per_layer
    for _ in range(40):
        last = net.train_step(x, labels, 2e-2, True)
    assert last < 0.6 * loss, (last, loss)
    # inference uses the running statistics of both kinds of batch norm; the class map has one entry per image
    cls = net.infer(x[0])
    assert cls.shape == (1, 1, 1) and 0 <= cls[0, 0, 0] < classes


def test_none_can_mean_no_activation(monkeypatch):
    """NONE on a CONV layer is ReLU, as in the reference (convolution_layer_functions.h:59-65); with
    UNET_NONE_IS_IDENTITY=1 it means what it says.  One hidden layer whose batch-norm output is half negative."""
    import neural_net as nn
    graph = [("conv", [8, 8, 3, 4, 3, 1, 1], [], "relu"), ("conv", [8, 8, 4, 1, 3, 1, 1], [0], "sigmoid")]
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 3, 8, 8)).astype(np.float32)
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("UNET_NONE_IS_IDENTITY", flag)
        net = nn.NeuralNet()
        net.set_seed(9)
        net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
        net.forward(x, False)
        outs[flag] = net.layer_output(0, 2)
    assert outs["0"].min() == 0.0                       # ReLU
    assert outs["1"].min() < -0.5                       # the batch norm's output itself
    np.testing.assert_array_equal(np.maximum(outs["1"], 0), outs["0"])
