#!/bin/bash
# Profiles judged under profiles/ (run on the GPU box:  gpurun -- bash tools/collect_profiles.sh r03):
#   * kernel-time stats of the benchmark command for every configuration (cfg2 = headline; cfg1, cfg3, cfg4 bf16 / fp8,
#     cfg5; cfg2 in fp32 -- the precision at which the north star's 1e-3 logit tolerance holds);
#   * separate PMC passes (never combined with a trace domain): FETCH_SIZE and WRITE_SIZE per configuration, folded by
#     tools/pmc_traffic.py into HBM bytes per launch and per step; one SQ / GRBM pass for cfg2 (MFMA-pipe utilisation).
set -o pipefail
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
prof=$out/final   # (only gpurun_out/ travels back from the GPU box: copy prof_$tag/final/* into profiles/ afterwards)
mkdir -p $out $out/final
cd /tmp && export TMPDIR=/tmp
run_stats() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$name -- python3 $root/bench.py "$@" \
    --steps 12 --warmup 3 --no-cpu-baseline > $out/bench_$name.json 2> $out/bench_$name.err || echo "stats $name failed"
  find $out/stats_$name -name "*kernel_trace.csv" -delete
  local st=$(find $out/stats_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$st" ] && cp $st $prof/${tag}_bench_kernel_stats_$name.csv
  grep '^{' $out/bench_$name.json > $prof/${tag}_bench_under_rocprof_$name.json || true
  echo "[collect] stats $name done"
}
run_pmc() {     # name, config, dtype, bench args...
  local name=$1 cfg=$2 dt=$3; shift 3
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$name -- python3 $root/bench.py "$@" --steps 2 --warmup 1 \
    --no-cpu-baseline --no-gemm-trace > /dev/null 2> $out/pmc_fetch_$name.err || echo "pmc fetch $name failed"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$name -- python3 $root/bench.py "$@" --steps 2 --warmup 1 \
    --no-cpu-baseline --no-gemm-trace > /dev/null 2> $out/pmc_write_$name.err || echo "pmc write $name failed"
  python3 $root/tools/pmc_traffic.py $out/pmc_fetch_$name $out/pmc_write_$name $prof/${tag}_pmc_hbm_traffic_$name.json \
    $prof/${tag}_pmc_hbm_traffic_$name.txt $cfg $dt 3 || echo "fold $name failed"
  rm -rf $out/pmc_fetch_$name $out/pmc_write_$name
  echo "[collect] pmc $name done"
}
# PLAIN_ONLY=1: only the un-profiled lines (after the PMC files of this tag have been copied into profiles/: the lines
# then carry `traffic_stale: false` against them)
if [ -z "$PLAIN_ONLY" ]; then
run_stats cfg2 --config cfg2
run_pmc cfg2 cfg2 bf16 --config cfg2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --output-format csv -d $out/pmc_sq -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-trace \
  > /dev/null 2> $out/pmc_sq.err || echo "pmc sq failed"
python3 $root/tools/pmc_mfma.py $out/pmc_sq $prof/${tag}_pmc_bench_sq.txt > /dev/null 2>&1 || echo "sq fold failed"
rm -rf $out/pmc_sq
echo "[collect] sq done"
run_stats cfg2_fp32 --config cfg2 --dtype fp32
run_stats cfg4_bf16 --config cfg4
run_pmc cfg4_bf16 cfg4 bf16 --config cfg4
run_stats cfg4_fp8 --config cfg4 --dtype fp8
run_pmc cfg4_fp8 cfg4 fp8 --config cfg4 --dtype fp8
run_stats cfg1 --config cfg1
run_pmc cfg1 cfg1 bf16 --config cfg1
run_stats cfg3 --config cfg3
run_stats cfg3_slic --config cfg3 --slic
run_pmc cfg3 cfg3 bf16 --config cfg3
run_stats cfg5 --config cfg5
run_pmc cfg5 cfg5 bf16 --config cfg5
fi
# un-profiled bench lines of every configuration on this box (the default command first, with its CPU baseline)
run_plain() {   # name, bench args...
  local name=$1; shift
  python3 $root/bench.py "$@" > $out/plain_$name.json 2> $out/plain_$name.err || echo "plain $name failed"
  grep '^{' $out/plain_$name.json > $prof/${tag}_bench_unprofiled_$name.json || true
  echo "[collect] plain $name done"
}
cd $root
python3 $root/bench.py > $out/plain_default.json 2> $out/plain_default.err || echo "default run failed"
grep '^{' $out/plain_default.json > $prof/${tag}_bench_default_run.json || true
echo "[collect] default run done"
run_plain cfg2 --config cfg2 --no-cpu-baseline
run_plain cfg2_drop --config cfg2 --dropout 0.1 --no-cpu-baseline
run_plain cfg2_drop_all --config cfg2 --dropout 0.1 --attn-dropout 0.1 --embed-dropout 0.1 --no-cpu-baseline
run_plain cfg1 --config cfg1 --no-cpu-baseline
run_plain cfg3 --config cfg3 --no-cpu-baseline
run_plain cfg3_slic --config cfg3 --slic --no-cpu-baseline
run_plain cfg5 --config cfg5 --no-cpu-baseline
run_plain cfg4_bf16 --config cfg4 --no-cpu-baseline
run_plain cfg4_fp8 --config cfg4 --dtype fp8 --no-cpu-baseline
run_plain cfg2_fp32 --config cfg2 --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline
ls -la $prof | tail -60
