#!/bin/bash
# A/B of one environment switch on ONE box: alternates `bench.py` runs without / with the variable set and prints ms per step.
#   tools/ab_bench.sh TCVOM_NO_GEMM_PAIR [rounds=3] [bench args...]
var=$1; rounds=${2:-3}; shift; shift
case "$var" in *=*) ;; *) var="$var=1";; esac      # NAME (-> NAME=1) or NAME=VALUE
for i in $(seq $rounds); do
  a=$(python bench.py --steps 12 --no-cpu-baseline --no-profile "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env $var python bench.py --steps 12 --no-cpu-baseline --no-profile "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "default $a ms   $var $b ms"
done
