#!/usr/bin/env python3
"""What a training step costs while a second stream holds some workgroup slots (as the resident kernels of a
collective do during the gradient all-reduce): forward + backward + step of the default bench workload with
0, 8, 16, 32, 64 occupying workgroups, each sleeping without memory traffic for the length of a step.
usage: lab_occupied_cus.py [LDS bytes per occupying workgroup]   (an RCCL workgroup on gfx950 holds 37664)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UNET_QUIET", "1")
import numpy as np, torch
import neural_net as nn, torch_ref
from unet_amd import capi
lab = ctypes.CDLL(os.path.join(ROOT, "tools", "bin", "liboccupy_lab.so"))
lab.occupy.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda", 0); capi.unet_set_device(0)
graph = torch_ref.unet_graph(256, 256, 3, 64, 4, True)
net = nn.NeuralNet(); net.set_seed(1)
net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS); net.reserve(16)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((16, 3, 256, 256)).astype(np.float32)).to(dev)
m = torch.from_numpy((rng.random((16, 256, 256)) < 0.03).astype(np.float32)).to(dev)
side = ctypes.c_void_p(); capi.unet_stream_create(ctypes.byref(side))
lds = int(sys.argv[1]) if len(sys.argv) > 1 else 0
def run(wgs, steps=20):
    def one():
        if wgs:
            lab.occupy(side, wgs, 6000, lds)    # holds its slots for about one step
        net.forward_backward_device(x.data_ptr(), m.data_ptr(), 16)
        net.step(1e-3)
        net.synchronize(); capi.unet_stream_synchronize(side)
    for _ in range(3): one()
    t0 = time.perf_counter()
    for _ in range(steps): one()
    return (time.perf_counter() - t0) / steps * 1e3
for wgs in (0, 8, 16, 32, 64, 0):
    print(f"{wgs:3d} occupying workgroups (256 threads, {lds} B of LDS each): {run(wgs):.3f} ms per step "
          "(synchronised every step)")
