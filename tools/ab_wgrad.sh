set -e
mkdir -p gpurun_out/ab_wg
timeout -k 10 300 python -u -m pytest tests/test_ops_bf16_gpu.py -x -q -k "wgrad" --timeout 120 --timeout-method thread > gpurun_out/ab_wg/pytest.log 2>&1 || { tail -30 gpurun_out/ab_wg/pytest.log; exit 1; }
tail -3 gpurun_out/ab_wg/pytest.log
for i in 1 2; do for m in 0 1; do
  UNET_LAB_WGRAD_DMA=$m timeout -k 10 100 python bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/ab_wg/m${m}_$i.json
done; done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/ab_wg/*.json")):
    d = json.loads(open(f).read().strip().split("\n")[-1]); k = d["kernel_families"]
    print(f.split("/")[-1], "step", d["ms_per_step"], "wgrad", k["conv3x3_wgrad_bf16"]["ms_per_step"], "reduce", k.get("wgrad_reduce",{}).get("ms_per_step"))
PY
