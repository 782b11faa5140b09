#!/usr/bin/env python3
"""Throughput benchmark of the U-Net training hot path.

    python bench.py --gpus N --steps K --warmup W

A step = one forward + backward + Adam update over one batch of synthetic
images on every rank, one process per GPU.  For N > 1 the ranks come either from
a launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N
...`, which sets WORLD_SIZE) or, when the script is started plainly, from the
script itself: it starts that launcher as a child process, relays its output and
exits with its status (and with an error if the line the ranks print does not say
n_gpus = N).  `--dry-launch` shows the ranks that would run and touches no GPU.
Inputs are resident in HBM before the timed region starts.  Rank 0 prints ONE
JSON line: whole-job images/s, the roofline position of the dominant kernel
family (timed live with HIP events on the launch stream) and, at N = 1, a CPU
baseline of a bounded sample of the same workload.

Workload (config.workload): BASELINE.json configs[2] -- U-Net fwd+bwd training
step with BatchNorm, batch 16 x 3 x 256 x 256 per GPU -- on the classic
4-down/4-up double-conv base-64 graph of configs[1] ("unet4_base64") or the
reference's own main.py graph ("mainpy", 3 down / 3 up, single conv).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UNET_QUIET", "1")

# Dense peaks used for roofline fractions (MI355X, gfx950):
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # bf16 MFMA dense; fp32 vector/matrix
PEAK_HBM_GBS = 8000.0


def conv_flops(graph, batch):
    """Algorithmic FLOPs of one training step: 2*N*Co*Ho*Wo*Ci*k*k per conv pass,
    three passes (forward, data gradient, weight gradient); the first layer has
    no data gradient."""
    fwd = 0.0
    total = 0.0
    for kind, s, parents, _ in graph:
        if kind == "conv":
            f = 2.0 * s[0] * s[1] * s[3] * s[2] * s[4] * s[4]
        elif kind == "up":
            f = 2.0 * (s[0] * s[4]) * (s[1] * s[4]) * s[3] * s[2]
        else:
            continue
        fwd += f
        total += f * (3 if parents else 2)
    return fwd * batch, total * batch


def family_flops(graph, batch):
    """Algorithmic FLOPs per step of each timed MFMA kernel family, from the layer table alone
    (logical channel counts: 3 input channels on the first layer, not the 32 the kernel pads to).
    The one-output-channel head goes to its own family."""
    fam = {"conv3x3_fwd_bf16": 0.0, "conv3x3_dgrad_bf16": 0.0, "conv3x3_wgrad_bf16": 0.0, "tconv2x2_bf16": 0.0,
           "tconv2x2_wgrad_bf16": 0.0, "conv3x3_head_bf16": 0.0}
    for kind, s, parents, _ in graph:
        if kind == "conv":
            f = 2.0 * s[0] * s[1] * s[3] * s[2] * s[4] * s[4] * batch
            if s[3] == 1:
                fam["conv3x3_head_bf16"] += f * (3 if parents else 2)
                continue
            fam["conv3x3_fwd_bf16"] += f
            fam["conv3x3_wgrad_bf16"] += f
            if parents:
                fam["conv3x3_dgrad_bf16"] += f
        elif kind == "up":
            f = 2.0 * (s[0] * s[4]) * (s[1] * s[4]) * s[3] * s[2] * batch
            fam["tconv2x2_bf16"] += 2 * f      # forward and data gradient run the same kernel
            fam["tconv2x2_wgrad_bf16"] += f
    return fam


def source_hash(*names):
    """Short hash of kernel source files: ties committed counter data to the build it was measured on."""
    import hashlib
    h = hashlib.sha1()
    for n in names:
        with open(os.path.join(ROOT, "unet_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def cpu_baseline_port(seconds_budget=30.0):
    """Times the CPU oracle (single thread, plain C) on a bounded sample of the
    workload: three images through the 64->64 3x3 convolution at 256x256 (forward +
    backward), i.e. one of the two stage-1 convolutions; scales by FLOPs to an
    images/s figure for the whole step."""
    import oracle
    rng = np.random.default_rng(0)
    hw = 256
    x = rng.standard_normal((3, 64, hw, hw)).astype(np.float32)
    w = (rng.standard_normal(64 * 64 * 9 + 64) * 0.05).astype(np.float32)
    t0 = time.perf_counter()
    y = oracle.convolve(x, w, 64, 3, 1, 1)
    dy = rng.standard_normal(y.shape).astype(np.float32)
    oracle.convolve_backward(x, dy, w, 3, 1, 1)
    dt = time.perf_counter() - t0
    return dt, 3 * 3 * 2.0 * hw * hw * 64 * 64 * 9


def cpu_baseline_torch(graph, batch=1, steps=1):
    """PyTorch-CPU (the oracle of test/torch_compare.py): fwd+bwd of the same
    graph in fp32 with all host threads."""
    import torch
    import torch_ref
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    n = torch_ref.param_count(graph)
    flat = rng.standard_normal(n) * 0.05
    params = torch_ref.split_params(graph, flat)
    for p in params:
        if p is not None:
            for k in p:
                p[k] = p[k].detach().float().requires_grad_(True)
    s0 = graph[0][1]
    x = torch.randn(batch, s0[2], s0[0], s0[1])
    m = (torch.rand(batch, s0[0], s0[1]) < 0.03).float()
    t0 = time.perf_counter()
    for _ in range(steps):
        acts = torch_ref.forward(graph, params, x)
        torch_ref.dice_loss(acts[-1], m).backward()
    return (time.perf_counter() - t0) / steps, cores


def emit_line(rec):
    """One JSON record as ONE write of the whole line, newline included: the ranks of a launcher share a pipe,
    and a record written apart from its newline can end up on the same line as another rank's."""
    sys.stdout.flush()
    sys.stdout.write(json.dumps(rec) + "\n")
    sys.stdout.flush()


def launch_ranks(args):
    """Started without a launcher but asked for several GPUs: run the same command line under
    torch.distributed.run as a CHILD process (one rank per GPU, rendezvous on 127.0.0.1), relay what it
    prints, and fail unless rank 0's line reports the requested number of GPUs."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, UNET_BENCH_SELF_LAUNCHED="1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    reported, dry_ranks = None, set()
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        text = line.strip()
        if text.startswith("{"):
            try:
                rec = json.loads(text)
            except ValueError:
                continue
            if "n_gpus" in rec:
                reported = rec["n_gpus"]
            if "dry_launch_rank" in rec:
                dry_ranks.add(rec["dry_launch_rank"])
    rc = proc.wait()
    if rc == 0 and args.dry_launch and dry_ranks != set(range(args.gpus)):
        print(f"bench.py: --gpus {args.gpus} but the ranks that started were {sorted(dry_ranks)}", file=sys.stderr)
        rc = 1
    if rc == 0 and not args.dry_launch and reported != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the ranks reported n_gpus = {reported}", file=sys.stderr)
        rc = 1
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=int(os.environ.get("UNET_NGPU", "1") or 1),
                    help="GPUs of this node, one rank each (default: UNET_NGPU, else 1)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the ranks, have each print its RANK / WORLD_SIZE, touch no GPU")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--base", type=int, default=64, help="channels of the first stage (unet4 graph)")
    ap.add_argument("--graph", default=os.environ.get("UNET_BENCH_GRAPH", "unet4_base64"),
                    choices=["unet4_base64", "mainpy", "attention_unet"])
    ap.add_argument("--loss", default="dice", choices=["dice", "xent"],
                    help="dice: the reference's loss on a one-channel sigmoid output; xent: softmax cross-entropy over "
                         "--classes logits (the final layer gets that many channels)")
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--infer", action="store_true",
                    help="time the inference forward pass (BASELINE config 2 with --dtype fp32 --batch 1) instead of a train step")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32"], help="compute path (default: bf16 where the graph allows)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the per-kernel event timing")
    args = ap.parse_args()

    # Ranks first, before anything touches a GPU: a plain `python bench.py --gpus N` starts its own launcher.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_launch:
        emit_line({"dry_launch_rank": int(os.environ.get("RANK", "0")),
                   "world_size": int(os.environ.get("WORLD_SIZE", "1")),
                   "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                   "master": f"{os.environ.get('MASTER_ADDR', '-')}:{os.environ.get('MASTER_PORT', '-')}"})
        return
    if args.dtype:
        os.environ["UNET_DTYPE"] = args.dtype
    import torch
    import torch.distributed as dist
    import neural_net as nn
    import torch_ref
    from unet_amd import capi, dp

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    capi.unet_set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    H = W = args.size
    if args.graph == "unet4_base64":
        graph = torch_ref.unet_graph(H, W, 3, args.base, 4, True)
    elif args.graph == "mainpy":
        graph = torch_ref.main_py_graph(H, W, 3, 64)
    else:   # the graph main.py:144-184 keeps commented out: main.py's graph with an attention gate on every skip
        graph = torch_ref.attention_unet_graph(H, W, 3, 64)
    free0, _ = torch.cuda.mem_get_info(device)
    net = nn.NeuralNet()
    net.set_seed(1)
    if args.loss == "xent":   # the final layer's channels become the classes' logits
        kind, shape, parents, _ = graph[-1]
        graph[-1] = (kind, shape[:3] + [args.classes] + shape[4:], parents, "logits")
        descs = torch_ref.to_layer_descs(nn, graph[:-1])
        descs.append(nn.LayerDesc(nn.LayerType.CONV_LAYER, graph[-1][1], parents, nn.ActivationType.SOFTMAX))
        net.create(descs, nn.LossFunction.CROSS_ENTROPY_LOSS)
    else:
        net.create(torch_ref.to_layer_descs(nn, graph), nn.LossFunction.DICE_LOSS)
    net.reserve(args.batch)
    dtype = net.compute_dtype() if hasattr(net, "compute_dtype") else "f32"

    # synthetic data, generated once, resident on the device
    rng = np.random.default_rng(1234 + rank)
    images = torch.from_numpy(rng.standard_normal((args.batch, 3, H, W)).astype(np.float32)).to(device)
    if args.loss == "xent":
        masks = torch.from_numpy(rng.integers(0, args.classes, (args.batch, H, W)).astype(np.float32)).to(device)
    else:
        masks = torch.from_numpy((rng.random((args.batch, H, W)) < 0.03).astype(np.float32)).to(device)
    free1, _ = torch.cuda.mem_get_info(device)   # engine buffers + the synthetic batch
    trainer = dp.DataParallel(net, device)
    trainer.broadcast_parameters()
    lr = 1e-3
    # What carries the gradient mean, as the communicator itself reports it (the line must show whether a
    # multi-GPU figure came from the overlapped RCCL path of the C ABI or from a fallback).
    rccl_ranks, transport = 0, "none (one rank)"
    if trainer.comm is not None:
        ws, rk = ctypes.c_int(), ctypes.c_int()
        capi.unet_comm_size(trainer.comm, ctypes.byref(ws), ctypes.byref(rk))
        rccl_ranks, transport = ws.value, "rccl-native"
    elif trainer.torch_device_reduce:
        transport = "torch-fallback"
    elif world > 1:
        transport = "host-staged"

    def one_step():
        if args.infer:
            net.forward_device(images.data_ptr(), args.batch, True)
        else:
            trainer.train_step(images.data_ptr(), masks.data_ptr(), args.batch, lr)

    def sync():
        net.synchronize()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    # Timer families -> kernels.  The forward and data-gradient 3x3 convolutions are the same kernels (the
    # data gradient runs them on flipped weights): conv3x3_pipe_kernel for images of at least 32 x 16 pixels
    # with enough tiles to fill the device, conv_mfma_kernel<9,64> for the rest (16 x 16 layers here).
    CONV3 = "conv3x3_pipe_kernel+conv_mfma_kernel<9,64>"
    # 3x3 weight gradient: the direct-to-LDS kernel; the register-staged one for images of 2^30 bytes and more
    WGRAD3 = "wgrad_dma_kernel<9,3>+wgrad_mfma_kernel<9>"
    # transposed convolution, forward and data gradient: one matrix product per launch for channel counts in
    # multiples of 64 (every layer of this graph), conv_mfma_kernel<1,64> for the rest
    TCONV = "tconv_gemm_kernel+conv_mfma_kernel<1,64>"
    kernel_of = {"conv3x3_fwd_bf16": CONV3, "conv3x3_dgrad_bf16": CONV3,
                 "conv3x3_wgrad_bf16": WGRAD3, "tconv2x2_bf16": TCONV,
                 "tconv2x2_wgrad_bf16": "wgrad_mfma_kernel<1,4>"}
    # substrings of the rocprofv3 kernel names that belong to a group (for the counter-derived traffic)
    trace_names = {CONV3: ("conv3x3_pipe_kernel<", "conv_mfma_kernel<9, 64,"),
                   WGRAD3: ("wgrad_dma_kernel<9", "wgrad_mfma_kernel<9>"),
                   TCONV: ("tconv_gemm_kernel<", "conv_mfma_kernel<1, 64,"), "wgrad_mfma_kernel<1,4>": ("wgrad_mfma_kernel<1, 4>", "wgrad_mfma_kernel<1>")}

    def read_families():
        fams = {}
        kinds = ctypes.c_int()
        capi.unet_prof_kinds(ctypes.byref(kinds))
        for k in range(kinds.value):
            n, ms, fl, by = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            capi.unet_prof_get(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
            if n.value:
                fams[capi.lib().unet_prof_name(k).decode()] = dict(
                    kind=k, launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value)
        return fams

    def algorithmic(fams, steps):
        """The library's timers attribute FLOPs from the padded channel counts its kernels see; replace
        them by the algorithmic figure of the layer table for the families the table covers."""
        alg = family_flops(graph, args.batch)
        for name, f in fams.items():
            if name in alg and dtype == "bf16" and not args.infer:
                f["flops"] = alg[name] * steps
        return fams

    def group_kernels(fams):
        kernels = {}
        for fam, f in fams.items():
            k = kernels.setdefault(kernel_of.get(fam, fam), dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, mask=0))
            for key in ("launches", "ms", "flops", "bytes"):
                k[key] += f[key]
            k["mask"] |= 1 << f["kind"]
        return kernels

    for _ in range(args.warmup):
        one_step()
    sync()

    # Untimed probe: every kernel family timed with HIP events (two event records per launch cost
    # about 1 ms per step, so this pass is kept out of the timed region).  It yields the per-family
    # table and names the dominant kernel.
    probe_steps = 0 if args.no_prof else 3
    probe, dominant_mask = {}, 0
    if probe_steps:
        capi.unet_prof_enable(1)
        for _ in range(probe_steps):
            one_step()
        sync()
        probe = algorithmic(read_families(), probe_steps)
        dominant_mask = max(group_kernels(probe).values(), key=lambda k: k["ms"])["mask"]
        # Timed region: events only around the launches of the dominant kernel.
        capi.unet_prof_enable_kinds(dominant_mask)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    families = {}
    if probe_steps:
        families = algorithmic(read_families(), args.steps)   # the dominant kernel's families, timed inside the timed region
        capi.unet_prof_enable(0)

    loss = None
    if rank == 0 and not args.infer and args.loss == "xent":
        lv = torch.zeros(1, device=device)
        capi.unet_memcpy_d2d(lv.data_ptr(), net.loss_values_ptr(), 4, None)
        torch.cuda.synchronize(device)
        loss = float(lv.item())
    elif rank == 0 and not args.infer:
        lv = torch.zeros(3 * args.batch, device=device)
        capi.unet_memcpy_d2d(lv.data_ptr(), net.loss_values_ptr(), 12 * args.batch, None)
        torch.cuda.synchronize(device)
        lv = lv.cpu().numpy().reshape(-1, 3)
        loss = float(1.0 - np.mean(2 * lv[:, 0] / (lv[:, 1] + lv[:, 2] + 1e-5)))

    trainer.close()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = args.batch * world * args.steps / elapsed
    fwd_flops, step_flops = conv_flops(graph, args.batch)
    out = {
        "metric": (f"U-Net inference images/s (fwd, {H}x{W}, {dtype})" if args.infer
                   else f"U-Net train images/s (fwd+bwd, {H}x{W}, {dtype})"),
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"{args.graph if args.base == 64 else 'unet4_base%d' % args.base} "
                               f"{'inference forward (BatchNorm with running statistics)' if args.infer else ('train step (fwd+bwd+Adam, BatchNorm, Dice loss)' if args.loss == 'dice' else f'train step (fwd+bwd+Adam, BatchNorm, softmax cross-entropy over {args.classes} classes)')}, "
                               f"batch {args.batch}x3x{H}x{W} per GPU",
                   "global_batch": args.batch * world, "parallelism": f"dp{world}",
                   "parameters": int(net.num_parameters()),
                   "conv_tflop_per_step": round((fwd_flops if args.infer else step_flops) / 1e12, 4)},
        "hbm_in_use_gb": round((free0 - free1) / 1e9, 2),
        "step_tflops": round((fwd_flops if args.infer else step_flops) * world / elapsed * args.steps / 1e12, 2),
        "final_loss": loss,
        "rccl_ranks": rccl_ranks, "transport": transport,
        "gradient_dtype_on_links": ("bf16" if os.environ.get("UNET_GRAD_BF16", "0") not in ("", "0") else "f32") if rccl_ranks else None,
    }
    if families:
        mfma_bound = {CONV3, WGRAD3, TCONV,
                      "wgrad_mfma_kernel<1,4>", "conv_fwd_f32", "conv_dgrad_f32", "conv_wgrad_f32", "tconv_f32"}
        # dominant kernel = the one with the largest share of device time in the probe pass; its
        # launches were timed live in the timed region
        (name, f), = group_kernels(families).items()
        if name in mfma_bound:
            achieved = f["flops"] / (f["ms"] * 1e-3) / 1e12
            peak = PEAK_TFLOPS["f32" if name.endswith("f32") else "bf16"]
            out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak,
                               "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
                               "avg_launch_us": round(f["ms"] * 1e3 / f["launches"], 2), "launches": f["launches"]}
        else:
            achieved = f["bytes"] / (f["ms"] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": None,
                               "avg_launch_us": round(f["ms"] * 1e3 / f["launches"], 2), "launches": f["launches"]}
        # HBM-side bytes per launch of that kernel, measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # (two passes of this same command, summarised by tools/summarize_pmc.py) -- only valid for
        # the default workload the counters were collected on.
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic_per_kernel.json")
        default_workload = (args.graph == "unet4_base64" and args.batch == 16 and args.size == 256 and
                            args.base == 64 and dtype == "bf16")
        traffic = json.load(open(tpath)) if default_workload and os.path.exists(tpath) else {}
        # the counters describe one build of the kernels: a file collected on other sources is not reported
        if traffic.get("_source_hash") == source_hash("conv_bf16.hip", "conv3x3_pipe.h", "wgrad_bf16.hip", "bf16_util.h"):
            # launch-weighted mean over the template instances of the group (with / without the fused
            # batch-norm-backward reduction, resident weights or not)
            recs = [rec for kname, rec in traffic.items()
                    if isinstance(rec, dict) and any(t in kname for t in trace_names.get(name, (name,)))]
            if recs:
                n = sum(r["launches"] for r in recs)
                out["roofline"]["traffic"] = round(sum(r["traffic_bytes_per_launch"] * r["launches"] for r in recs) / n)
                out["roofline"]["algorithmic_bytes_per_launch"] = round(f["bytes"] / f["launches"])
        # per-family table from the untimed probe pass (all families timed)
        out["kernel_families"] = {
            k: {"launches_per_step": v["launches"] // probe_steps, "ms_per_step": round(v["ms"] / probe_steps, 3),
                "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None}
            for k, v in sorted(probe.items(), key=lambda kv: -kv[1]["ms"])}
    if world == 1 and not args.no_cpu_baseline and not args.infer:
        dt, sample_flops = cpu_baseline_port()
        img_flops = step_flops / args.batch
        out["cpu_baseline"] = {
            "value": round(1.0 / (dt * img_flops / sample_flops), 5), "unit": "images/s", "cores": 1,
            "kind": "port",
            "sample": f"C oracle, 3 images through one 64->64 3x3 conv at 256x256 fwd+bwd ({dt:.1f} s, "
                      f"{sample_flops / 1e9:.1f} GFLOP of {img_flops / 1e9:.0f} GFLOP/image), scaled by FLOPs"}
        try:
            tdt, cores = cpu_baseline_torch(graph, 1, 1)
            out["cpu_torch_oracle"] = {"value": round(1.0 / tdt, 4), "unit": "images/s", "cores": cores,
                                       "sample": f"PyTorch-CPU fp32 fwd+bwd of the same graph, batch 1, {tdt:.1f} s"}
        except Exception as e:  # the torch figure is a side note; never fail the bench on it
            out["cpu_torch_oracle"] = {"error": str(e)[:200]}
    emit_line(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
