"""Two identically seeded networks, one training step each on the same batch: lists the layers whose
gradients differ bit-wise (none expected)."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.environ.get("UNET_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import neural_net as nn  # noqa: E402
import torch_ref  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
graph = torch_ref.unet_graph(H, H, 3, 64, 4, True)
rng = np.random.default_rng(22)
n = 16
x = rng.standard_normal((n, 3, H, H)).astype(np.float32)
m = (rng.random((n, H, H)) < 0.03).astype(np.float32)
nets = []
for _ in range(2):
    net = nn.NeuralNet()
    net.set_seed(3)
    net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
    nets.append(net)
g = []
for net in nets:
    net.train_step(x, m, 1e-3, False)
    g.append(net.get_gradients())
for rep in range(2):   # and the same network again
    nets[0].train_step(x, m, 1e-3, False)
    g.append(nets[0].get_gradients())
layout = nets[0].parameter_layout()
bad = 0
for ent in layout:
    i, off, cnt = int(ent[0]), int(ent[1]), int(ent[2]) + int(ent[3])
    eq = [np.array_equal(g[0][off:off + cnt], gi[off:off + cnt]) for gi in g[1:]]
    if not all(eq):
        bad += 1
        d = max(np.abs(g[0][off:off + cnt] - gi[off:off + cnt]).max() for gi in g[1:])
        print(f"layer {i}: differs {eq} max abs diff {d:.3e} (max |g| {np.abs(g[0][off:off+cnt]).max():.3e})")
print("layers with differences:", bad)
