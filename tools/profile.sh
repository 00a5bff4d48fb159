#!/bin/sh
# tools/profile.sh -- rocprofv3 passes for one bench workload (run on the GPU box).
# usage: tools/profile.sh <workload> <outdir> [extra bench args]
# Pass 1: --kernel-trace --stats (per-kernel time).  Passes 2,3: PMC FETCH_SIZE / WRITE_SIZE, each on
# its own (TCC slots: FETCH_SIZE 3 + WRITE_SIZE 2 do not fit one pass; never mixed with sys/hip traces).
WL=$1; OUT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
FSM_BENCH_DETAIL="$OUT/bench_detail_trace.json" rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python "$ROOT/bench.py" --workload "$WL" --steps 5 --warmup 2 --no-cpu-baseline --subs none "$@" > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
FSM_BENCH_DETAIL="$OUT/bench_detail_fetch.json" rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- python "$ROOT/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline --subs none "$@" > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
FSM_BENCH_DETAIL="$OUT/bench_detail_write.json" rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o write -- python "$ROOT/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline --subs none "$@" > "$OUT/bench_write.json" 2> "$OUT/write.err"
# pass 4: the fabric read requests BY SIZE (gfx950 exposes the split): bytes read = 32 a + 64 b + 128 c, with no assumption about
# the kernel's access pattern (FETCH_SIZE tallies every request as 64 bytes)
FSM_BENCH_DETAIL="$OUT/bench_detail_req.json" rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum -d "$OUT/req" -o req -- python "$ROOT/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline --subs none "$@" > "$OUT/bench_req.json" 2> "$OUT/req.err"
# optional pass 5 (L2_PMC="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"): L2 requests / hits / misses, for the gather-bound workloads
if [ -n "$L2_PMC" ]; then
	FSM_BENCH_DETAIL="$OUT/bench_detail_l2.json" rocprofv3 --pmc $L2_PMC -d "$OUT/l2" -o l2 -- python "$ROOT/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline --subs none "$@" > "$OUT/bench_l2.json" 2> "$OUT/l2.err"
fi
find "$OUT" -name "*.db" | head -20
