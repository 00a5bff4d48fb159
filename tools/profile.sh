#!/bin/sh
# tools/profile.sh -- rocprofv3 passes for one bench workload (run on the GPU box).
# usage: tools/profile.sh <workload> <outdir> [extra bench args]
# Pass 1: --kernel-trace --stats (per-kernel time).  Passes 2,3: PMC FETCH_SIZE / WRITE_SIZE, each on
# its own (TCC slots: FETCH_SIZE 3 + WRITE_SIZE 2 do not fit one pass; never mixed with sys/hip traces).
WL=$1; OUT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python "$ROOT/bench.py" --workload "$WL" --steps 5 --warmup 2 --no-cpu-baseline --subs none "$@" > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- python "$ROOT/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline --subs none "$@" > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o write -- python "$ROOT/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline --subs none "$@" > "$OUT/bench_write.json" 2> "$OUT/write.err"
# optional pass 4 (L2_PMC="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"): L2 requests / hits / misses, for the gather-bound workloads
if [ -n "$L2_PMC" ]; then
	rocprofv3 --pmc $L2_PMC -d "$OUT/l2" -o l2 -- python "$ROOT/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline --subs none "$@" > "$OUT/bench_l2.json" 2> "$OUT/l2.err"
fi
find "$OUT" -name "*.db" | head -20
