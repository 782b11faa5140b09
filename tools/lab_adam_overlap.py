#!/usr/bin/env python3
"""Does an HBM-bound Adam pass hide under the MFMA-bound kernels of a training step?  Times, on the default
bench workload: (a) forward + backward alone, (b) + the engine's Adam and weight repack in series (the normal
step), (c) forward + backward with an Adam over scratch vectors of the same size enqueued on a second stream at
the start of every step.  (c) - (a) is what an overlapped optimiser would cost."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UNET_QUIET", "1")
import numpy as np, torch
import neural_net as nn, torch_ref
from unet_amd import capi
dev = torch.device("cuda", 0); capi.unet_set_device(0)
graph = torch_ref.unet_graph(256, 256, 3, 64, 4, True)
net = nn.NeuralNet(); net.set_seed(1)
net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS); net.reserve(16)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((16, 3, 256, 256)).astype(np.float32)).to(dev)
m = torch.from_numpy((rng.random((16, 256, 256)) < 0.03).astype(np.float32)).to(dev)
n = net.num_parameters()
w2, g2, s2 = torch.zeros(n, device=dev), torch.randn(n, device=dev) * 1e-3, torch.zeros(2 * n, device=dev)
side = ctypes.c_void_p(); capi.unet_stream_create(ctypes.byref(side))
def run(kind, steps=30):
    def one(t):
        if kind == "side":
            capi.unet_update_weights_adam_f32(w2.data_ptr(), g2.data_ptr(), s2.data_ptr(), 1e-3, n, t + 1, side)
        net.forward_backward_device(x.data_ptr(), m.data_ptr(), 16)
        if kind == "serial":
            net.step(1e-3)
    for t in range(5): one(t)
    net.synchronize(); capi.unet_stream_synchronize(side)
    t0 = time.perf_counter()
    for t in range(steps): one(t)
    net.synchronize(); capi.unet_stream_synchronize(side)
    return (time.perf_counter() - t0) / steps * 1e3
for rep in range(2):
    print("fwd+bwd %.3f ms | + Adam and repack in series %.3f ms | + Adam on a second stream %.3f ms" %
          (run("none"), run("serial"), run("side")))
