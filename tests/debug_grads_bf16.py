import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["UNET_QUIET"] = "1"
import numpy as np, torch, torch_ref
import neural_net as nn
graph = torch_ref.main_py_graph(32, 48, 3, 32)
def make(dtype=None):
    if dtype: os.environ["UNET_DTYPE"] = dtype
    else: os.environ.pop("UNET_DTYPE", None)
    net = nn.NeuralNet(); net.set_seed(3)
    net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
    return net
rng = np.random.default_rng(7)
n = 4
x = rng.standard_normal((n, 3, 32, 48)).astype(np.float32)
m = (rng.random((n, 32, 48)) < 0.3).astype(np.float32)
net = make()
p0 = net.get_parameters()
p0 = (p0 + rng.standard_normal(p0.size).astype(np.float32) * 0.02).astype(np.float32)
net.set_parameters(p0)
net.train_step(x, m, 1e-2, False)
g16 = net.get_gradients()
net32 = make("fp32"); net32.set_parameters(p0); net32.train_step(x, m, 1e-2, False); g32 = net32.get_gradients()
params = torch_ref.split_params(graph, p0)
acts = torch_ref.forward(graph, params, torch.from_numpy(x).double(), emulate_bf16=True)
torch_ref.dice_loss(acts[-1], torch.from_numpy(m).double()).backward()
ref = torch_ref.flat_grads(graph, params)
for i in range(len(graph)):
    a = net.layer_output(i, n); r = acts[i].detach().numpy()
    print("act", i, graph[i][0], "rel L2 %.3g  max %.3g  nonequal frac %.3g" % (np.linalg.norm(a-r)/np.linalg.norm(r), np.abs(a-r).max(), (a!=r).mean()))
for layer, off, nwb, nbn in net.parameter_layout():
    co = nbn // 2
    for lo, hi, what in ((off, off + nwb - co, "w"), (off + nwb - co, off + nwb, "bias"), (off + nwb, off + nwb + nbn, "gb")):
        r = ref[lo:hi]; a = g16[lo:hi]; b = g32[lo:hi]
        print(layer, what, "bf16 rel %.3g  fp32 rel %.3g  |ref| %.3g" % (np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-30), np.linalg.norm(b - r) / max(np.linalg.norm(r), 1e-30), np.linalg.norm(r)))
