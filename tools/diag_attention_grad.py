#!/usr/bin/env python3
"""Per-layer gradient error of the bf16 Attention U-Net test graph against the bf16-storage oracle (the numbers
tests/test_attention_net_gpu.py bounds), for comparing two builds: run from the root of either tree."""
import os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UNET_QUIET", "1")
import numpy as np, torch
import neural_net as nn, torch_ref
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
graph = torch_ref.attention_unet_graph(32, 48, 3, 32)
net = nn.NeuralNet(); net.set_seed(1)
net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
rng = np.random.default_rng(seed)
n = 4
x = rng.standard_normal((n, 3, 32, 48)).astype(np.float32)
m = (rng.random((n, 32, 48)) < 0.3).astype(np.float32)
p0 = net.get_parameters()
p0 = (p0 + rng.standard_normal(p0.size).astype(np.float32) * 0.02).astype(np.float32)
net.set_parameters(p0)
net.train_step(x, m, 1e-2, False)
grads = net.get_gradients()
params = torch_ref.split_params(graph, p0)
acts = torch_ref.forward(graph, params, torch.from_numpy(x).double(), emulate_bf16=True)
torch_ref.dice_loss(acts[-1], torch.from_numpy(m).double()).backward()
ref = torch_ref.flat_grads(graph, params)
out = []
for layer, off, nwb, nbn in net.parameter_layout():
    co = nbn // 2
    for lo, hi, what in [(off, off + nwb - co, "w")] + ([(off + nwb, off + nwb + nbn, "gb")] if nbn else []):
        g, r = grads[lo:hi].astype(np.float64), ref[lo:hi]
        out.append(f"{layer}{what}:{np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-12):.3f}")
print(f"seed {seed}:", " ".join(out))
