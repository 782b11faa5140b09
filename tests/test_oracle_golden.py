"""Pins the CPU oracle (oracle/unet_oracle.c) against the PyTorch-CPU golden
vectors in tests/golden/ops_small.npz (made by tests/golden/make_golden.py).
The reference itself holds no golden vectors for this path; PyTorch-CPU is the
oracle its test/torch_compare.py names.  CPU only."""
import numpy as np
import pytest

from util import assert_close, assert_exact

CONV = ["conv_a", "conv_b", "conv_c", "conv_d", "conv_e", "conv_f", "conv_g"]


@pytest.mark.parametrize("name", CONV)
def test_conv_forward(oracle, golden, name):
    n, ci, co, h, w, k, p, s = golden[name + "_cfg"]
    y = oracle.convolve(golden[name + "_x"], golden[name + "_w"], int(co), int(k), int(p), int(s))
    assert_close(y, golden[name + "_y"], what=name)


@pytest.mark.parametrize("name", CONV)
def test_conv_backward(oracle, golden, name):
    n, ci, co, h, w, k, p, s = golden[name + "_cfg"]
    dx, dw = oracle.convolve_backward(golden[name + "_x"], golden[name + "_dy"], golden[name + "_w"],
                                      int(k), int(p), int(s))
    assert_close(dx, golden[name + "_dx"], what=name + " dx")
    assert_close(dw, golden[name + "_dw"], what=name + " dw")


@pytest.mark.parametrize("name", ["bn_a", "bn_b"])
def test_batch_norm(oracle, golden, name):
    x, gb = golden[name + "_x"], golden[name + "_gb"]
    y, stats, run = oracle.bn_train(x, gb, golden[name + "_run_in"])
    assert_close(y, golden[name + "_y"], what=name + " y")
    assert_close(stats, golden[name + "_stats"], what=name + " raw sums")
    assert_close(run, golden[name + "_run_out"], what=name + " running stats")
    dx, dgb = oracle.bn_backward(x, golden[name + "_dy"], gb, stats)
    assert_close(dx, golden[name + "_dx"], rtol=2e-4, what=name + " dx")
    assert_close(dgb, golden[name + "_dgb"], what=name + " dgamma/dbeta")
    yi = oracle.bn_inference(x, gb, golden[name + "_run_in"])
    assert_close(yi, golden[name + "_yinf"], what=name + " inference")


@pytest.mark.parametrize("name", ["pool_a", "pool_b"])
def test_max_pool(oracle, golden, name):
    x = golden[name + "_x"]
    y, idx, tm = oracle.maxpool(x, 2, 2, want_indices=True, want_tiemask=True)
    assert_exact(y, golden[name + "_y"], what=name + " y")
    assert_exact(idx, golden[name + "_idx"], what=name + " indices")
    assert_exact(tm, golden[name + "_tiemask"], what=name + " tie mask")
    dy = golden[name + "_dy"]
    assert_exact(oracle.maxpool_backward(dy, x, y, 2), golden[name + "_dx_ties"], what=name + " dx (ties)")
    assert_exact(oracle.maxpool_backward_indices(dy, idx, x.shape[2], x.shape[3]),
                 golden[name + "_dx_first"], what=name + " dx (indices)")


@pytest.mark.parametrize("name", ["up_a", "up_b", "up_c"])
def test_upsample(oracle, golden, name):
    n, ci, co, h, w, s = golden[name + "_cfg"]
    x, wt = golden[name + "_x"], golden[name + "_w"]
    assert_close(oracle.upsample(x, wt, int(co), int(s)), golden[name + "_y"], what=name)
    dx, dw = oracle.upsample_backward(x, wt, golden[name + "_dy"], int(s))
    assert_close(dx, golden[name + "_dx"], what=name + " dx")
    assert_close(dw, golden[name + "_dw"], what=name + " dw")


def test_concat_layout(oracle, golden):
    a, b = golden["cat_a"], golden["cat_b"]
    y = oracle.concat(a, b)
    assert_exact(y, golden["cat_y"], what="concat")
    da, db = oracle.concat_backward(y, a.shape[1])
    assert_exact(da, a, what="concat backward A")
    assert_exact(db, b, what="concat backward B")


def test_activations(oracle, golden):
    x, dy = golden["act_x"], golden["act_dy"]
    assert_exact(oracle.relu(x), golden["relu_y"], what="relu")
    assert_exact(oracle.relu_backward(dy, x), golden["relu_dx"], what="relu backward")
    y = oracle.sigmoid(x)
    assert_close(y, golden["sigm_y"], what="sigmoid")
    assert_close(oracle.sigmoid_backward(dy, y), golden["sigm_dx"], what="sigmoid backward")


def test_losses(oracle, golden):
    d, lv = oracle.dice_score(golden["dice_mask"], golden["dice_pred"])
    assert_close(lv, golden["dice_lv"], what="dice sums")
    assert_close(np.float32(d), golden["dice_score"], what="dice score")
    assert_close(oracle.dice_backward(golden["dice_mask"], lv), golden["dice_grad_ref"], what="dice backward")
    assert_close(oracle.difference_backward(golden["diff_m"], golden["diff_y"]), golden["diff_grad"],
                 what="difference loss backward")
    loss, g = oracle.softmax_xent(golden["xent_logits"], golden["xent_target"])
    assert_close(np.float32(loss), golden["xent_loss"], what="cross entropy")
    assert_close(g, golden["xent_grad"], what="cross entropy grad")


def test_adam(oracle, golden):
    w = golden["adam_w0"]
    st = np.zeros(2 * w.size, np.float32)
    for t, g in enumerate(golden["adam_grads"], start=1):
        w, st = oracle.adam(w, g, st, 0.01, t)
    assert_close(w, golden["adam_w3"], what="adam weights")
    assert_close(st, golden["adam_state3"], rtol=1e-4, what="adam state")


def test_small_ops(oracle):
    rng = np.random.default_rng(0)
    a = rng.standard_normal(100).astype(np.float32)
    b = rng.standard_normal(100).astype(np.float32)
    assert_exact(oracle.add_b_to_a(a, b), a + b)
    assert_exact(oracle.threshold(a, 0.5), (a >= 0.5).astype(np.float32))
    assert_exact(oracle.sgd(a, b, 0.1), a - b * np.float32(0.1))


def test_dice_reference_formula_is_exact_derivative(golden):
    """Documents that the reference's backward formula (given the sums) IS the
    exact per-sample derivative of the Dice score -- autograd agrees."""
    assert_close(golden["dice_grad_ref"], golden["dice_grad_full"], rtol=1e-4, what="dice derivative")


ATTENTION = ["attn_a", "attn_odd", "attn_b"]


@pytest.mark.parametrize("name", ATTENTION)
def test_attention_gate(oracle, golden_attention, name):
    """tests/golden/attention_small.npz (tests/golden/make_golden_attention.py): torch operators + autograd."""
    G = golden_attention
    ci = int(G[name + "_cfg"][3])
    x, g, params = G[name + "_x"], G[name + "_g"], G[name + "_params"]
    y, int1, alpha = oracle.attention(x, g, params, ci)
    assert_close(int1, G[name + "_int1"], what=name + " int1")
    assert_close(alpha, G[name + "_alpha"], what=name + " alpha")
    assert_close(y, G[name + "_y"], what=name + " y")
    dx, dg, dp = oracle.attention_backward(G[name + "_dy"], int1, alpha, x, g, params)
    assert_close(dx, G[name + "_dx"], what=name + " dx")
    assert_close(dg, G[name + "_dg"], what=name + " dg")
    assert_close(dp, G[name + "_dparams"], what=name + " dparams")
