mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -x -q --timeout 500 -k "lazy" > gpurun_out/r07f_pytest5.log 2>&1; tail -5 gpurun_out/r07f_pytest5.log
export FSM_BENCH_LINES_FORMS=off64
for wl in c5_ragged c5_short; do
    timeout 300 python bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', r['value'], 'GB/s', r['ms_per_step'], 'ms', r['roofline']['kernel'])"
done > gpurun_out/r07f_c5_lines.txt 2>&1
cat gpurun_out/r07f_c5_lines.txt
