#!/bin/bash
# A/B of two builds of the library on ONE box: alternates bench.py runs with the default library and with TCVOM_LIB=<other .so>.
#   tools/ab_lib.sh tcvom_amd/lib/libtcvom_hip_f16_base.so [rounds=3] [bench args...]
lib=$1; rounds=${2:-3}; shift; shift
for i in $(seq $rounds); do
  a=$(python bench.py --steps 12 --no-cpu-baseline --no-profile "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env TCVOM_LIB=$PWD/$lib python bench.py --steps 12 --no-cpu-baseline --no-profile "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "default $a ms   $lib $b ms"
done
