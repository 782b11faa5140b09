"""Checkpoints and PyTorch weight exchange for a created neural_net.NeuralNet.

The reference cannot persist a trained model (src/neural_net.h keeps everything
in device memory until exit); this module adds that on top of the module's
parameter getters/setters.

Parameter vector layout (the reference's per-layer order, src/layer.h:445-447):
per CONV / UPSAMPLING layer  [ W (C_in, C_out, k, k) | bias (C_out) | gamma0, beta0, gamma1, beta1, ... ].
Running statistics per layer: [mean0, var0, mean1, var1, ...].

  save(net, path) / load(net, path)         .npz with parameters, running statistics, Adam state
  to_torch(net)   -> {"layers.<i>.conv.weight": ..., ...}   tensors in PyTorch's layouts
  from_torch(net, state)                    the inverse

PyTorch layouts: nn.Conv2d weight is (C_out, C_in, k, k) -- transposed with
respect to the reference; nn.ConvTranspose2d weight is (C_in, C_out, k, k) --
the same; BatchNorm2d keeps weight (gamma), bias (beta), running_mean,
running_var as separate vectors.
"""
import os

import numpy as np

FORMAT_VERSION = 1


def _npz_path(path):
    """numpy.savez appends ".npz" to a name without that suffix; save() and load() agree on the result."""
    path = os.fspath(path)
    return path if path.endswith(".npz") else path + ".npz"


def _layers(net):
    """[(layer index, parameter offset, number of weights+biases, number of BN params, output shape)]"""
    shapes = net.output_shapes()
    return [(i, off, nwb, nbn, shapes[i]) for i, off, nwb, nbn in net.parameter_layout()]


def split_parameters(flat, layout):
    """flat vector -> {layer: dict(wb=..., gamma=..., beta=...)} (views into a copy)."""
    flat = np.asarray(flat, dtype=np.float32)
    out = {}
    for i, off, nwb, nbn in layout:
        bn = flat[off + nwb:off + nwb + nbn]
        out[i] = {"wb": flat[off:off + nwb].copy(), "gamma": bn[0::2].copy(), "beta": bn[1::2].copy()}
    return out


def save(net, path, include_optimizer=True):
    """Writes parameters, running statistics and (optionally) the Adam state to `path` (.npz)."""
    layout = net.parameter_layout()
    arrays = {
        "format_version": np.int64(FORMAT_VERSION),
        "parameters": np.asarray(net.get_parameters(), dtype=np.float32),
        "layout": np.asarray(layout, dtype=np.int64).reshape(-1, 4),
    }
    for i, _, _, nbn in layout:
        if nbn:
            arrays[f"running_{i}"] = np.asarray(net.running_stats(i), dtype=np.float32)
    if include_optimizer:
        arrays["adam_state"] = np.asarray(net.get_optimizer_state(), dtype=np.float32)
        arrays["adam_steps"] = np.int64(net.optimizer_steps())
    with open(_npz_path(path), "wb") as fh:
        np.savez(fh, **arrays)


def load(net, path, include_optimizer=True):
    """Restores what save() wrote into a network created from the same layer list."""
    with np.load(_npz_path(path), allow_pickle=False) as f:
        if int(f["format_version"]) != FORMAT_VERSION:
            raise RuntimeError(f"unsupported checkpoint version {int(f['format_version'])}")
        layout = np.asarray(net.parameter_layout(), dtype=np.int64).reshape(-1, 4)
        if f["layout"].shape != layout.shape or not np.array_equal(f["layout"], layout):
            raise RuntimeError("the checkpoint was written by a network with a different layer list")
        # every array is read and checked against the network before the first one is copied in, so that a
        # truncated or edited file leaves the network as it was
        n_params = int(net.num_parameters())
        params = np.ascontiguousarray(f["parameters"], dtype=np.float32)
        if params.size != n_params:
            raise RuntimeError(f"checkpoint holds {params.size} parameters, the network has {n_params}")
        running = {}
        for i, _, _, nbn in layout:
            if nbn:
                key = f"running_{int(i)}"
                if key not in f.files:
                    raise RuntimeError(f"checkpoint has no running statistics for layer {int(i)}")
                running[int(i)] = np.ascontiguousarray(f[key], dtype=np.float32)
                if running[int(i)].size != int(nbn):
                    raise RuntimeError(f"checkpoint: running statistics of layer {int(i)} have "
                                       f"{running[int(i)].size} entries, expected {int(nbn)}")
        adam = None
        if include_optimizer and "adam_state" in f.files:
            adam = np.ascontiguousarray(f["adam_state"], dtype=np.float32)
            if adam.size != 2 * n_params or "adam_steps" not in f.files:
                raise RuntimeError(f"checkpoint: Adam state has {adam.size} entries, expected {2 * n_params}")
            adam_steps = int(f["adam_steps"])
    net.set_parameters(params)
    for i, r in running.items():
        net.set_running_stats(i, r)
    if adam is not None:
        net.set_optimizer_state(adam, adam_steps)


def layer_to_torch(wb, c_in, c_out, k, transposed):
    """[W (C_in,C_out,k,k) | bias] -> (weight in PyTorch's layout, bias)."""
    nw = c_in * c_out * k * k
    w = np.asarray(wb[:nw], dtype=np.float32).reshape(c_in, c_out, k, k)
    if not transposed:              # nn.Conv2d wants (C_out, C_in, k, k)
        w = w.transpose(1, 0, 2, 3)
    return np.ascontiguousarray(w), np.asarray(wb[nw:nw + c_out], dtype=np.float32).copy()


def layer_from_torch(weight, bias, transposed):
    """PyTorch (weight, bias) -> [W (C_in,C_out,k,k) | bias] flat."""
    w = np.asarray(weight, dtype=np.float32)
    if not transposed:
        w = w.transpose(1, 0, 2, 3)
    return np.concatenate([np.ascontiguousarray(w).reshape(-1), np.asarray(bias, dtype=np.float32).reshape(-1)])


ATTENTION_KEYS = ("w_x.weight", "w_g.weight", "w_g.bias", "psi.weight", "psi.bias")


def attention_to_torch(flat, c_x, c_g):
    """Attention-gate parameters W[(Cx+Cg)][Ci] | psi[Ci] | b1[Ci] | bpsi -> the tensors of the usual PyTorch
    attention gate: w_x = Conv2d(Cx, Ci, 1, stride=2, bias=False), w_g = Conv2d(Cg, Ci, 1), psi = Conv2d(Ci, 1, 1)."""
    flat = np.asarray(flat, dtype=np.float32)
    ci = (flat.size - 1) // (c_x + c_g + 2)
    if (c_x + c_g + 2) * ci + 1 != flat.size:
        raise RuntimeError(f"attention gate: {flat.size} parameters do not fit Cx={c_x}, Cg={c_g}")
    w = flat[:(c_x + c_g) * ci].reshape(c_x + c_g, ci)
    k = (c_x + c_g) * ci
    return {"w_x.weight": np.ascontiguousarray(w[:c_x].T).reshape(ci, c_x, 1, 1),
            "w_g.weight": np.ascontiguousarray(w[c_x:].T).reshape(ci, c_g, 1, 1),
            "w_g.bias": flat[k + ci:k + 2 * ci].copy(),
            "psi.weight": flat[k:k + ci].reshape(1, ci, 1, 1).copy(),
            "psi.bias": flat[k + 2 * ci:k + 2 * ci + 1].copy()}


def attention_from_torch(t):
    """Inverse of attention_to_torch()."""
    wx, wg = np.asarray(t["w_x.weight"], np.float32), np.asarray(t["w_g.weight"], np.float32)
    ci = wx.shape[0]
    w = np.concatenate([wx.reshape(ci, -1).T, wg.reshape(ci, -1).T], axis=0)
    return np.concatenate([w.reshape(-1), np.asarray(t["psi.weight"], np.float32).reshape(-1),
                           np.asarray(t["w_g.bias"], np.float32).reshape(-1),
                           np.asarray(t["psi.bias"], np.float32).reshape(-1)])


def _kind(desc):
    """'conv', 'up', 'attention', 'bn' or 'fc' for a LayerDesc with parameters."""
    name = getattr(desc.type, "name", str(desc.type))
    return {"CONV_LAYER": "conv", "UPSAMPLING_LAYER": "up", "ATTENTION_LAYER": "attention", "BN_LAYER": "bn",
            "FC_LAYER": "fc"}[name.split(".")[-1]]


def to_torch(net, descs):
    """descs: the LayerDesc list the network was created from.  Returns numpy arrays keyed like a
    PyTorch state_dict: layers.<i>.conv.{weight,bias}, layers.<i>.bn.{weight,bias,running_mean,running_var};
    a BN_LAYER has the bn entries only, an FC_LAYER layers.<i>.fc.{weight [out, in], bias} (torch.nn.Linear)."""
    flat = np.asarray(net.get_parameters(), dtype=np.float32)
    parts = split_parameters(flat, net.parameter_layout())
    out = {}
    shapes = net.output_shapes()
    for i, off, nwb, nbn in net.parameter_layout():
        kind, p = _kind(descs[i]), parts[i]
        if kind == "attention":                 # parents = [x, g]
            cx, cg = (shapes[q][0] for q in descs[i].parents)
            for key, v in attention_to_torch(flat[off:off + nwb], cx, cg).items():
                out[f"layers.{i}.{key}"] = v
            continue
        shape = list(descs[i].shape)
        if kind == "fc":                        # [C_in, H_in, W_in, out]: weights [in][out], then the biases
            n_out = shape[3]
            out[f"layers.{i}.fc.weight"] = p["wb"][:nwb - n_out].reshape(-1, n_out).T.copy()
            out[f"layers.{i}.fc.bias"] = p["wb"][nwb - n_out:].copy()
            continue
        if kind in ("conv", "up"):
            transposed = kind == "up"           # UPSAMPLING: [H, W, C_in, C_out, scale]
            c_in, c_out, k = shape[2], shape[3], shape[4]
            w, b = layer_to_torch(p["wb"], c_in, c_out, k, transposed)
            out[f"layers.{i}.conv.weight"] = w
            out[f"layers.{i}.conv.bias"] = b
        r = np.asarray(net.running_stats(i), dtype=np.float32)
        out[f"layers.{i}.bn.weight"] = p["gamma"]
        out[f"layers.{i}.bn.bias"] = p["beta"]
        out[f"layers.{i}.bn.running_mean"] = r[0::2].copy()
        out[f"layers.{i}.bn.running_var"] = r[1::2].copy()
    return out


def from_torch(net, descs, state):
    """Inverse of to_torch(); `state` maps the same keys to arrays / tensors."""
    def arr(v):
        return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
    flat = np.asarray(net.get_parameters(), dtype=np.float32).copy()
    for i, off, nwb, nbn in net.parameter_layout():
        kind = _kind(descs[i])
        if kind == "attention":
            wb = attention_from_torch({k: arr(state[f"layers.{i}.{k}"]) for k in ATTENTION_KEYS})
            if wb.size != nwb:
                raise RuntimeError(f"layer {i}: expected {nwb} attention parameters, got {wb.size}")
            flat[off:off + nwb] = wb
            continue
        if kind == "fc":
            wb = np.concatenate([arr(state[f"layers.{i}.fc.weight"]).T.reshape(-1), arr(state[f"layers.{i}.fc.bias"]).reshape(-1)])
            if wb.size != nwb:
                raise RuntimeError(f"layer {i}: expected {nwb} weights and biases, got {wb.size}")
            flat[off:off + nwb] = wb.astype(np.float32)
            continue
        if kind in ("conv", "up"):
            wb = layer_from_torch(arr(state[f"layers.{i}.conv.weight"]), arr(state[f"layers.{i}.conv.bias"]), kind == "up")
            if wb.size != nwb:
                raise RuntimeError(f"layer {i}: expected {nwb} weights and biases, got {wb.size}")
            flat[off:off + nwb] = wb
        bn = np.empty(nbn, np.float32)
        bn[0::2] = arr(state[f"layers.{i}.bn.weight"])
        bn[1::2] = arr(state[f"layers.{i}.bn.bias"])
        flat[off + nwb:off + nwb + nbn] = bn
        r = np.empty(nbn, np.float32)
        r[0::2] = arr(state[f"layers.{i}.bn.running_mean"])
        r[1::2] = arr(state[f"layers.{i}.bn.running_var"])
        net.set_running_stats(int(i), r)
    net.set_parameters(flat)
