#!/bin/bash
# stagger experiment (probe build): does delaying one of a CU's two first-round workgroups overlap epilogues with main loops?
cd "$(dirname "$0")/.."
for dbg in 64 $((16+5*256)) $((16+10*256)) $((16+15*256)) $((16+20*256)) $((48+5*256)) $((48+10*256)) $((48+15*256)) $((48+20*256)); do
  echo "== FAVIT_GEMM_DBG=$dbg (delay $((dbg>>8)) us, mode $(((dbg>>5)&1)))"
  FAVIT_GEMM_DBG=$dbg REPS=30 timeout -k 10 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
