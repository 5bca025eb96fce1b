#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call (box-to-box variance is ~5 %):
#   tools/ab.sh ab/base.so [ab/other.so ...]   -> ms_per_step of each, interleaved, 3 rounds
set -e
for r in 1 2 3; do
  for lib in "$@" mpc_amd/libmpc_amd.so; do
    MPC_AMD_LIB=$lib timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-e2e \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib', d['ms_per_step'])"
  done
done
