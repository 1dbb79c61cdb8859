#!/bin/bash
# usage: ab.sh ENVVAR "v1 v2" reps [bench args]
var=$1; vals=$2; reps=$3; shift 3
for r in $(seq $reps); do for v in $vals; do
  ms=$(env $var=$v timeout -k 10 200 python bench.py --steps 4 --no-cpu-baseline --no-roofline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$var=$v $* : $ms"
done; done
