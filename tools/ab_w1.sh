#!/bin/bash
# A/B of the one-stage-per-lane kernels (batch = 1 latency, forced wave1 at the bench size, full SQP)
for r in 1 2; do
  for lib in "$@" mpc_amd/libmpc_amd.so; do
    MPC_AMD_LIB=$lib MPC_MODEL=wave1 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib wave1 ms', round(d['ms_per_step'],3), 'b1 p50', round(d['latency_b1_ms']['p50'],4), 'kernel', round(d['latency_b1_ms']['kernel_p50'],4))"
  done
done
for lib in "$@" mpc_amd/libmpc_amd.so; do
  MPC_AMD_LIB=$lib timeout -k 10 200 python tools/scratch/sqp_split.py 2>&1 | grep "^wall" | cut -c1-60 | sed "s|^|$lib |"
done
