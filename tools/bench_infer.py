"""Inference throughput of the unet4_base64 graph (forward with running statistics) on device-resident input:
images/s at batch 1 and 16.  Not the headline metric; shows what folding BN + ReLU into the conv epilogue buys."""
import os
import sys
import time
import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.environ.get("UNET_ROOT", ROOT))
import neural_net as nn  # noqa: E402
import torch_ref  # noqa: E402

dev = torch.device("cuda", 0)
graph = torch_ref.unet_graph(256, 256, 3, 64, 4, True)
net = nn.NeuralNet()
net.set_seed(1)
net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
for batch in (1, 16):
    net.reserve(batch)
    x = torch.randn(batch, 3, 256, 256, device=dev)
    for _ in range(5):
        net.forward_device(x.data_ptr(), batch, True)
    net.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        net.forward_device(x.data_ptr(), batch, True)
    net.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"inference batch {batch}: {dt * 1e3:.3f} ms per batch, {batch / dt:.1f} images/s")
