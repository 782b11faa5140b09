#!/bin/bash
# Runs on the GPU box (gpurun): the bench lines and rocprofv3 summaries that profiles/ keeps.
#   gpurun -- 'bash tools/collect_profiles.sh'      then, here:  bash tools/collect_profiles.sh --import
# Every GPU step has its own time limit; the script stops at the first failure.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_r3
if [ "$1" = "--import" ]; then
  P=$R/profiles
  cp $O/bench_default.json $P/r3_bench_default.json
  cp $O/bench_default_under_rocprof.json $P/r3_bench_default_under_rocprof.json
  cp $O/trace/bench_kernel_stats.csv $P/r3_bench_default_kernel_stats.csv
  python $R/tools/trace_by_shape.py $O/trace/bench_kernel_trace.csv 0 > $P/r3_bench_default_last_step_by_shape.txt
  cp $O/pmc_fetch/bench_counter_collection.csv $P/r3_pmc_fetch_size_counter_collection.csv
  cp $O/pmc_write/bench_counter_collection.csv $P/r3_pmc_write_size_counter_collection.csv
  python $R/tools/summarize_pmc.py $O/pmc_fetch/bench_counter_collection.csv $O/pmc_write/bench_counter_collection.csv $P/hbm_traffic_per_kernel.json
  cp $P/hbm_traffic_per_kernel.json $P/r3_hbm_traffic_per_kernel.json
  for f in config2_fp32_infer_b1 config2_fp32_infer_b16 config2_bf16_infer_b1 config2_bf16_infer_b16; do cp $O/$f.json $P/r3_$f.json; done
  cp $O/trace_cfg2/bench_kernel_stats.csv $P/r3_config2_fp32_infer_b1_kernel_stats.csv
  cp $O/pmc_sq1/bench_counter_collection.csv $P/r3_pmc_sq_mfma_counter_collection.csv
  cp $O/pmc_sq2/bench_counter_collection.csv $P/r3_pmc_sq_lds_counter_collection.csv
  python $R/tools/summarize_sq.py $O/pmc_sq1/bench_counter_collection.csv $O/pmc_sq2/bench_counter_collection.csv > $P/r3_sq_counters_per_kernel.md
  ls -la $P | tail -20
  exit 0
fi
mkdir -p $O
cd $R
if [ "$1" != "--sq-only" ]; then
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
fi
cd /tmp && export TMPDIR=/tmp
# SQ / GRBM passes (at most 8 SQ and 2 GRBM counters a pass; the same counters in both passes where a ratio needs them)
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq1 -o bench -- python3 $R/bench.py --no-cpu-baseline --no-prof --steps 3 --warmup 1 > $O/pmc_sq1.json 2> $O/pmc_sq1.err
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o bench -- python3 $R/bench.py --no-cpu-baseline --no-prof --steps 3 --warmup 1 > $O/pmc_sq2.json 2> $O/pmc_sq2.err
ls $O $O/trace $O/pmc_fetch $O/pmc_sq1
