"""The byte-input feed on the GPU: raw uint8 [N,H,W,C] images and uint8 masks, normalised on the device, must
train and infer like the reference's recipe (host-side /255 and z-score, main.py:85-98, then `train`)."""
import numpy as np
import pytest
import torch

import torch_ref
from unet_amd import data

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nn():
    import neural_net
    return neural_net


def test_normalize_kernels_match_numpy():
    import gpu_util as G
    from unet_amd import capi
    rng = np.random.default_rng(0)
    n, h, w, c = 3, 19, 23, 3
    img = rng.integers(0, 256, size=(n, h, w, c), dtype=np.uint8)
    scale = (rng.random(c) * 0.02 + 0.001).astype(np.float32)
    shift = rng.standard_normal(c).astype(np.float32)
    d_img = torch.from_numpy(img).to(G.DEV)
    out = G.empty((n, c, h, w))
    d_scale, d_shift = G.dev(scale), G.dev(shift)
    capi.unet_normalize_u8_nhwc_to_nchw_f32(d_img.data_ptr(), G.p(out), G.p(d_scale), G.p(d_shift), n, c, h, w, None)
    ref = np.transpose(img, (0, 3, 1, 2)).astype(np.float32) * scale.reshape(1, c, 1, 1) + shift.reshape(1, c, 1, 1)
    np.testing.assert_allclose(G.host(out), ref, rtol=1e-6, atol=1e-6)
    m = rng.integers(0, 2, size=(n, h, w), dtype=np.uint8) * 255
    outm = G.empty((n, h, w))
    d_m = torch.from_numpy(m).to(G.DEV)
    capi.unet_u8_to_f32(d_m.data_ptr(), G.p(outm), m.size, 1.0 / 255.0, None)
    np.testing.assert_array_equal(G.host(outm), (m > 0).astype(np.float32))


def test_train_u8_matches_host_normalised_training(nn):
    graph = torch_ref.main_py_graph(32, 32, 3, 32)
    images, masks = data.synthetic_dataset(10, 32, 32, seed=4)       # ragged last batch at batch size 4
    mean, std = data.channel_stats(images)
    nets = []
    for _ in range(2):
        net = nn.NeuralNet()
        net.set_seed(5)
        net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
        nets.append(net)
    a, b = nets
    np.testing.assert_array_equal(a.get_parameters(), b.get_parameters())
    with pytest.raises(RuntimeError):
        a.train_u8(images, masks, 1e-3, 1, 4)                          # normalisation not set yet
    a.set_input_normalization(mean, std)
    a.train_u8(images, masks, 1e-3, 2, 4)
    b.train(data.normalize_host(images, mean, std), data.masks_host(masks), 1e-3, 2, 4)
    pa, pb = a.get_parameters(), b.get_parameters()
    rel = np.linalg.norm(pa - pb) / np.linalg.norm(pb)
    assert rel <= 1e-2, rel
    assert abs(a.last_epoch_loss() - b.last_epoch_loss()) <= 2e-2
    ya = a.infer_u8(images[0])
    yb = a.infer(data.normalize_host(images[:1], mean, std)[0])
    assert ya.shape == (1, 32, 32)
    assert np.mean(ya != yb) <= 0.01
