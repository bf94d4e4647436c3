#!/bin/bash
# probe build: the step's GEMMs complete / main loop only / epilogue only, at two and at one workgroup per CU --
# does a phase's time per workgroup depend on its neighbour (throughput-bound) or not (latency-bound)?
cd "$(dirname "$0")/.."
for one in "" 1; do
  for dbg in 64 1 2; do
    echo "== one_per_cu=${one:-0} FAVIT_GEMM_DBG=$dbg (64: complete, 1: no epilogue, 2: no main loop)"
    if [ -n "$one" ]; then export FAVIT_GEMM_P4_ONE_PER_CU=1; else unset FAVIT_GEMM_P4_ONE_PER_CU; fi
    FAVIT_GEMM_DBG=$dbg REPS=30 timeout -k 10 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
  done
done
