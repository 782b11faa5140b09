#!/usr/bin/env python3
"""Throughput of NeuralNet.train() as main.py calls it: host numpy arrays in, every batch copied to the
device (the PCIe-inclusive rate; bench.py's figure starts from device-resident inputs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UNET_QUIET", "1")
import numpy as np, neural_net as nn, torch_ref
# usage: bench_train_api.py [unet4_base64|mainpy] [batch]   (main.py trains its graph with batches of 4)
which = sys.argv[1] if len(sys.argv) > 1 else "unet4_base64"
H, N, B = 256, 128, int(sys.argv[2]) if len(sys.argv) > 2 else 16
graph = torch_ref.main_py_graph(H, H, 3, 64) if which == "mainpy" else torch_ref.unet_graph(H, H, 3, 64, 4, True)
net = nn.NeuralNet(); net.set_seed(1)
net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
rng = np.random.default_rng(0)
x = rng.standard_normal((N, 3, H, H)).astype(np.float32)
m = (rng.random((N, H, H)) < 0.03).astype(np.float32)
net.train(x, m, 1e-3, 1, B)          # warm-up epoch
t0 = time.perf_counter()
epochs = 3
net.train(x, m, 1e-3, epochs, B)
dt = time.perf_counter() - t0
print(f"{which}: train() from host arrays: {N * epochs / dt:.1f} images/s ({dt / (epochs * N / B) * 1e3:.2f} ms per batch of {B}), loss {net.last_epoch_loss():.4f}")
