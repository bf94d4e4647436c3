#!/bin/bash
# Runs the given commands (one per argument) in order on the GPU box; an ordinary failure is recorded and the
# sequence continues, a timeout / kill (exit 124 / 137) ends it (no further GPU step after a hung one).
# usage: tools/gpu_seq.sh LOGDIR "timeout -k 10 300 cmd1 ..." "timeout -k 10 600 cmd2 ..."
dir=$1; shift
mkdir -p "$dir"
i=0
for c in "$@"; do
  i=$((i+1))
  echo "=== [$i] $c" | tee -a "$dir/seq.log"
  bash -c "$c" > "$dir/step$i.log" 2>&1
  rc=$?
  echo "=== [$i] rc=$rc" | tee -a "$dir/seq.log"
  tail -n 6 "$dir/step$i.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping" | tee -a "$dir/seq.log"; exit $rc; fi
done
exit 0
