#!/usr/bin/env python3
"""What the gradient all-reduce's resident kernels would cost a training step on a multi-GPU node, measured on ONE
GPU: the default bench workload with a one-rank communicator whose bucket all-reduce is replaced
(UNET_LAB_COMM_EMULATE, unet_amd/csrc/comm.hip) by a kernel that holds W workgroup slots with RCCL's LDS footprint
for bytes / rate seconds, issued exactly where the real collective is (per bucket, from backward, on the
communication stream).  One process per configuration (the switch is read once).

    python tools/lab_comm_overlap.py            # the table
    python tools/lab_comm_overlap.py --one W,LDS,GBPS [--bf16]   # one configuration (used by the table run)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UNET_QUIET", "1")


def one(spec, bf16):
    if spec != "none":
        os.environ["UNET_LAB_COMM_EMULATE"] = spec
    os.environ["UNET_COMM_FORCE"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    import numpy as np
    import torch
    import torch.distributed as dist
    import neural_net as nn
    import torch_ref
    from unet_amd import capi, dp
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    capi.unet_set_device(0)
    graph = torch_ref.unet_graph(256, 256, 3, 64, 4, True)
    net = nn.NeuralNet()
    net.set_seed(1)
    net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
    net.reserve(16)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((16, 3, 256, 256)).astype(np.float32)).to(dev)
    m = torch.from_numpy((rng.random((16, 256, 256)) < 0.03).astype(np.float32)).to(dev)
    trainer = None
    if spec != "none":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        trainer = dp.DataParallel(net, dev, force_collectives=True, bf16_gradients=bf16)
        assert trainer.comm is not None

    def step():
        net.forward_backward_device(x.data_ptr(), m.data_ptr(), 16)
        net.step(1e-3)

    for _ in range(5):
        step()
    net.synchronize()
    t0 = time.perf_counter()
    steps = 30
    for _ in range(steps):
        step()
    net.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"spec": spec, "bf16_gradients": bool(bf16), "ms_per_step": round(ms, 3),
                      "buckets": len(net.gradient_buckets())}), flush=True)
    if trainer:
        trainer.close()
        dist.destroy_process_group()


def table():
    rows = [("none", False)]
    for wgs in (16, 64):
        for gbps in (100, 200):
            rows.append((f"{wgs},37664,{gbps}", False))
    rows += [("64,37664,100", True), ("none", False)]
    base = None
    for spec, bf16 in rows:
        cmd = [sys.executable, os.path.abspath(__file__), "--one", spec] + (["--bf16"] if bf16 else [])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        recs = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
        if not recs:
            print(f"{spec}: failed\n{out.stderr[-400:]}")
            continue
        r = recs[-1]
        if spec == "none" and base is None:
            base = r["ms_per_step"]
        what = "no communicator" if spec == "none" else \
            "{} workgroups x {} B LDS, {} GB/s{}".format(*spec.split(","), ", bf16 buckets" if bf16 else "")
        print(f"{what:58s} {r['ms_per_step']:7.3f} ms/step  {100 * (r['ms_per_step'] / base - 1):+5.1f} %", flush=True)


if __name__ == "__main__":
    if "--one" in sys.argv:
        one(sys.argv[sys.argv.index("--one") + 1], "--bf16" in sys.argv)
    else:
        table()
