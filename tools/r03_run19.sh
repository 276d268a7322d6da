#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
bash tools/ab_bench.sh TCVOM_EXP_SHARED_W 3
