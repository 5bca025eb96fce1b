#!/bin/bash
# Profile the bench workload on the GPU box (run through gpurun): kernel trace + the PMC passes
# of profiles/*/README.md, each pass in its own rocprofv3 run (the MI355X guide: no counter
# multiplexing), then tools/prof_summary.py turns them into profiles-style CSV extracts + summary.json.
#   usage: bash tools/prof.sh <tag>     ->  gpurun_out/prof_<tag>/
set -u
TAG=${1:-run}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-latency --no-e2e --no-stress"
run() {  # name, rocprof args...
    local name=$1; shift
    timeout -s KILL 170 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
}
run trace --kernel-trace --stats || exit 1
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY || exit 1
run sqw --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VMEM SQ_WAVES || exit 1
run fetch --pmc FETCH_SIZE || exit 1
run write --pmc WRITE_SIZE || exit 1
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
