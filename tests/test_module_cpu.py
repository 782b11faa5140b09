"""CPU-only checks of the Python module surface (names the reference's
src/bindings/net.cpp:26-88 exports).  Constructing NeuralNet needs a GPU, so
only classes, enums and LayerDesc are exercised here."""
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
