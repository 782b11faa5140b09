import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["UNET_QUIET"] = "1"
import numpy as np, neural_net as nn, torch_ref
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
graph = torch_ref.unet_graph(H, H, 3, 64, 4, True)
def make(dtype):
    os.environ["UNET_DTYPE"] = dtype
    net = nn.NeuralNet(); net.set_seed(11)
    net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
    return net
rng = np.random.default_rng(22)
x = rng.standard_normal((n, 3, H, H)).astype(np.float32)
m = (rng.random((n, H, H)) < 0.03).astype(np.float32)
a = make("bf16"); f = make("fp32")
f.set_parameters(a.get_parameters())
la = a.train_step(x, m, 1e-3, False); lf = f.train_step(x, m, 1e-3, False)
ga = a.get_gradients().astype(np.float64); gf = f.get_gradients().astype(np.float64)
print("loss", la, lf)
off = 0
for i, (kind, s, _, _) in enumerate(graph):
    if kind in ("conv", "up"):
        ci, co, k = s[2], s[3], s[4]
        nw = ci * co * k * k
        for name, lo, hi in (("w", off, off + nw), ("b", off + nw, off + nw + co), ("gb", off + nw + co, off + nw + 3 * co)):
            u, v = ga[lo:hi], gf[lo:hi]
            c = np.dot(u, v) / max(np.linalg.norm(u) * np.linalg.norm(v), 1e-300)
            print(f"layer {i:2d} {kind:4s} {ci:4d}->{co:4d} {name:2s} cos={c:7.4f} |bf16|={np.linalg.norm(u):.3e} |fp32|={np.linalg.norm(v):.3e}")
        off += nw + 3 * co
