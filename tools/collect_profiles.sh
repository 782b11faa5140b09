#!/bin/bash
# Runs on the GPU box (gpurun): the bench lines and rocprofv3 summaries that profiles/ keeps.
#   gpurun -- 'bash tools/collect_profiles.sh'      then, here:  bash tools/collect_profiles.sh --import
# Every GPU step has its own time limit; the script stops at the first failure.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_r2
if [ "$1" = "--import" ]; then
  P=$R/profiles
  cp $O/bench_default.json $P/r2_bench_default.json
  cp $O/bench_default_under_rocprof.json $P/r2_bench_default_under_rocprof.json
  cp $O/trace/bench_kernel_stats.csv $P/r2_bench_default_kernel_stats.csv
  python $R/tools/trace_by_shape.py $O/trace/bench_kernel_trace.csv 0 > $P/r2_bench_default_last_step_by_shape.txt
  cp $O/pmc_fetch/bench_counter_collection.csv $P/r2_pmc_fetch_size_counter_collection.csv
  cp $O/pmc_write/bench_counter_collection.csv $P/r2_pmc_write_size_counter_collection.csv
  python $R/tools/summarize_pmc.py $O/pmc_fetch/bench_counter_collection.csv $O/pmc_write/bench_counter_collection.csv $P/hbm_traffic_per_kernel.json
  cp $P/hbm_traffic_per_kernel.json $P/r2_hbm_traffic_per_kernel.json
  for f in config2_fp32_infer_b1 config2_fp32_infer_b16 config2_bf16_infer_b1 config2_bf16_infer_b16; do cp $O/$f.json $P/r2_$f.json; done
  cp $O/trace_cfg2/bench_kernel_stats.csv $P/r2_config2_fp32_infer_b1_kernel_stats.csv
  ls -la $P | tail -20
  exit 0
fi
mkdir -p $O
cd $R
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for spec in "fp32 1" "fp32 16" "bf16 1" "bf16 16"; do
  set -- $spec
  timeout -k 10 200 python bench.py --infer --dtype $1 --batch $2 --steps 50 --warmup 10 > $O/config2_$1_infer_b$2.json 2> $O/config2.err
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench_default_under_rocprof.json 2> $O/trace.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg2 -o bench -- python3 $R/bench.py --infer --dtype fp32 --batch 1 --steps 50 --warmup 10 > $O/config2_fp32_infer_b1_under_rocprof.json 2> $O/trace2.err
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python3 $R/bench.py --no-cpu-baseline --no-prof --steps 3 --warmup 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python3 $R/bench.py --no-cpu-baseline --no-prof --steps 3 --warmup 1 > $O/pmc_write.json 2> $O/pmc_write.err
ls $O $O/trace $O/pmc_fetch
