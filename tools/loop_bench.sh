#!/bin/bash
# usage: loop.sh <n> <bench args...>  -> prints losses
n=$1; shift
bad=0
for i in $(seq 1 $n); do
  l=$(timeout -k 10 120 python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.readline()).get('loss'))")
  case "$l" in nan|NaN|inf|None|"") bad=$((bad+1)); echo "run $i: $l";; esac
done
echo "[$*] $n runs, $bad bad"
