#!/bin/bash
# Profiles judged under profiles/: kernel-time stats of the benchmark command for cfg2 (headline), cfg1 (dense
# attention) and cfg3 (SPPP), and separate PMC passes (HBM bytes, SQ wave states) for cfg2.  Run on the GPU box:
#   gpurun -- bash tools/collect_profiles.sh r02
set -o pipefail
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in cfg2 cfg1 cfg3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$cfg -- python3 $root/bench.py --config $cfg \
    --steps 12 --warmup 3 --no-cpu-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err || echo "stats $cfg failed"
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 2 --warmup 1 \
  --no-cpu-baseline --no-gemm-trace > /dev/null 2> $out/pmc_fetch.err || echo "pmc fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 2 --warmup 1 \
  --no-cpu-baseline --no-gemm-trace > /dev/null 2> $out/pmc_write.err || echo "pmc write failed"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --output-format csv -d $out/pmc_sq -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-trace \
  > /dev/null 2> $out/pmc_sq.err || echo "pmc sq failed"
find $out -name "*kernel_stats.csv" | head
