#!/bin/bash
# A/B harness for scheduling knobs: alternating bench.py runs under different environments (run it on the GPU box via gpurun).
# usage: ab2.sh "ENV1=a ENV2=b" "ENV1=c" ... -- [bench args]   (alternating runs, prints ms per cycle)
cfgs=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do cfgs+=("$1"); shift; done; shift
for r in 1 2; do for c in "${cfgs[@]}"; do
  ms=$(env $c timeout -k 10 200 python bench.py --steps 4 --no-cpu-baseline --no-roofline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$c $* : $ms"
done; done
