"""CPU-only checks of the Python module surface (names the reference's
src/bindings/net.cpp:26-88 exports).  Constructing NeuralNet needs a GPU, so
only classes, enums and LayerDesc are exercised here."""
import numpy as np

import neural_net as nn


def test_surface_names():
    for name in ["NeuralNet", "Convolution", "UNetTest", "LayerDesc", "LayerType", "ActivationType", "LossFunction"]:
        assert hasattr(nn, name), name
    for m in ["create", "train", "infer", "test"]:
        assert hasattr(nn.NeuralNet, m)
    for m in ["convolution2D", "train"]:
        assert hasattr(nn.Convolution, m)
    for m in ["testConvolution", "testUpsample", "testBatchNorm", "testMaxPool"]:
        assert hasattr(nn.UNetTest, m)
    # export_values(): LayerType and LossFunction members are module attributes, ActivationType's are not
    assert nn.CONV_LAYER == nn.LayerType.CONV_LAYER and nn.DICE_LOSS == nn.LossFunction.DICE_LOSS
    assert {"MAX_POOL_LAYER", "FC_LAYER", "CONV_LAYER", "ATTENTION_LAYER", "BN_LAYER", "UPSAMPLING_LAYER",
            "CONCAT_LAYER"} <= set(nn.LayerType.__members__)
    assert {"ReLU", "SOFTMAX", "NONE", "TANH", "SIGMOID", "LEAKY_ReLU"} == set(nn.ActivationType.__members__)
    assert not hasattr(nn, "SIGMOID")


def test_layer_desc():
    d = nn.LayerDesc(nn.LayerType.CONV_LAYER, [256, 256, 3, 64, 3, 1, 1], [])
    assert d.type == nn.LayerType.CONV_LAYER and d.shape == [256, 256, 3, 64, 3, 1, 1] and d.parents == []
    assert d.activation == nn.ActivationType.NONE and d.leaky_const == 0.01
    d = nn.LayerDesc(type=nn.LayerType.CONCAT_LAYER, shape=[], parents=[4, 7], activation=nn.ActivationType.SIGMOID,
                     leaky_const=0.2)
    d.parents = [1, 2]
    d.shape = [1]
    assert d.parents == [1, 2] and d.shape == [1] and d.leaky_const == 0.2


def test_network_level_reference_attention_gate_matches_the_oracle(oracle):
    """tests/torch_ref.py's attention gate (what the network-level GPU tests compare with) against the C oracle,
    forward and all gradients, on a shape with odd H/2 and W/2 (clamped blend cells)."""
    import torch
    import torch_ref
    rng = np.random.default_rng(21)
    n, cx, cg, ci, h, w = 2, 6, 5, 3, 10, 6
    x = rng.standard_normal((n, cx, h, w)).astype(np.float32)
    g = rng.standard_normal((n, cg, h // 2, w // 2)).astype(np.float32)
    flat = (rng.standard_normal(oracle.attention_params(cx, cg, ci)) * 0.3).astype(np.float32)
    dy = rng.standard_normal(x.shape).astype(np.float32)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    gt = torch.from_numpy(g).double().requires_grad_(True)
    pt = torch.from_numpy(flat).double().requires_grad_(True)
    y = torch_ref.attention_gate(xt, gt, pt, ci)
    y.backward(torch.from_numpy(dy).double())
    y_ref, int1, alpha = oracle.attention(x, g, flat, ci)
    dx, dg, dp = oracle.attention_backward(dy, int1, alpha, x, g, flat)
    np.testing.assert_allclose(y.detach().numpy(), y_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(xt.grad.numpy(), dx, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gt.grad.numpy(), dg, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(pt.grad.numpy(), dp, rtol=1e-4, atol=2e-5)
    graph = torch_ref.attention_unet_graph(16, 16, 3, 4)
    assert torch_ref.attention_dims(graph, 8) == (16, 32, 8) and torch_ref.attention_dims(graph, 16) == (4, 4, 2)
