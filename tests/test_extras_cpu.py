"""The oracle's restatements of the operators that have no loop in the reference's test headers and no
PyTorch golden file: LEAKY_ReLU / TANH (enum values of layer.h:31 that the reference runs as ReLU) and the
cropping concat (the extension that makes its unpadded-convolution descriptors usable).  Pinned here against
their closed forms in numpy / PyTorch float64 ("parity unpinned" by the reference itself, which has neither)."""
import numpy as np
import torch
import torch.nn.functional as F


def test_leaky_relu_and_tanh(oracle):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 5, 7, 6)).astype(np.float32)
    x[0, 0, 0, :3] = [0.0, -0.0, 1e-30]
    dy = rng.standard_normal(x.shape).astype(np.float32)
    for slope in (0.01, 0.2):
        y = oracle.leaky_relu(x, slope)
        np.testing.assert_array_equal(y, np.where(x > 0, x, np.float32(slope) * x))
        dx = oracle.leaky_relu_backward(dy, x, slope)
        np.testing.assert_array_equal(dx, np.where(x > 0, dy, np.float32(slope) * dy))
        t = torch.from_numpy(x).double().requires_grad_(True)
        F.leaky_relu(t, slope).backward(torch.from_numpy(dy).double())
        np.testing.assert_allclose(dx, t.grad.numpy(), rtol=1e-6, atol=1e-7)
    y = oracle.tanh(x)
    np.testing.assert_allclose(y, np.tanh(x.astype(np.float64)), rtol=1e-6, atol=1e-7)
    dx = oracle.tanh_backward(dy, y)
    np.testing.assert_allclose(dx, dy.astype(np.float64) * (1 - y.astype(np.float64) ** 2), rtol=1e-6, atol=1e-7)


def test_concat_crop(oracle):
    rng = np.random.default_rng(1)
    for (ha, wa, hb, wb) in [(12, 10, 8, 6), (9, 9, 9, 9), (7, 11, 5, 3)]:
        a = rng.standard_normal((2, 3, ha, wa)).astype(np.float32)
        b = rng.standard_normal((2, 4, hb, wb)).astype(np.float32)
        oy, ox = (ha - hb) // 2, (wa - wb) // 2
        want = np.concatenate([a[:, :, oy:oy + hb, ox:ox + wb], b], axis=1)
        np.testing.assert_array_equal(oracle.concat_crop(a, b), want)
        dy = rng.standard_normal(want.shape).astype(np.float32)
        da, db = oracle.concat_crop_backward(dy, 3, ha, wa)
        ta = torch.from_numpy(a).requires_grad_(True)
        tb = torch.from_numpy(b).requires_grad_(True)
        torch.cat([ta[:, :, oy:oy + hb, ox:ox + wb], tb], dim=1).backward(torch.from_numpy(dy))
        np.testing.assert_array_equal(da, ta.grad.numpy())
        np.testing.assert_array_equal(db, tb.grad.numpy())


def test_oracle_cross_entropy_ignores_labels_that_are_no_class():
    """oracle.softmax_xent against F.cross_entropy(ignore_index) in float64, every non-class label mapped to
    the ignored value on the torch side."""
    import torch
    import torch.nn.functional as F
    import oracle
    rng = np.random.default_rng(3)
    n, c, h, w = 2, 4, 9, 11
    logits = rng.standard_normal((n, c, h, w)).astype(np.float32) * 2
    tgt = rng.integers(0, c, (n, h, w)).astype(np.int32)
    tgt[rng.random(tgt.shape) < 0.25] = 255
    tgt[rng.random(tgt.shape) < 0.1] = -7
    x = torch.from_numpy(logits).double().requires_grad_(True)
    t = torch.from_numpy(np.where((tgt < 0) | (tgt >= c), -100, tgt)).long()
    ref = F.cross_entropy(x, t, ignore_index=-100)
    ref.backward()
    loss, g = oracle.softmax_xent(logits, tgt)
    assert abs(loss - float(ref)) <= 1e-5 * abs(float(ref))
    np.testing.assert_allclose(g, x.grad.numpy(), rtol=1e-4, atol=1e-7)
    loss0, g0 = oracle.softmax_xent(logits, np.full((n, h, w), -1, np.int32))
    assert loss0 == 0 and not g0.any()
