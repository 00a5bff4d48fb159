mkdir -p gpurun_out
export FSM_BENCH_LINES_FORMS=off64
for wl in c2_ragged c3_ragged; do
  for w in 0 10 8 6 4; do
    timeout 200 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --knob 4=$w 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl waves=$w', r['value'], 'GB/s', r['roofline']['kernel_ms_avg'], 'ms', r['roofline']['kernel'])"
  done
done > gpurun_out/r07l_ragged_waves.txt 2>&1
cat gpurun_out/r07l_ragged_waves.txt
cd /tmp; for w in 0 6; do rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d /tmp/rq$w -o r -- python $OLDPWD/bench.py --workload c2_ragged --steps 2 --warmup 1 --no-cpu-baseline --knob 4=$w > /dev/null 2>&1; python - <<PY
import sqlite3,glob
db=glob.glob('/tmp/rq$w/**/*.db',recursive=True)[0]
con=sqlite3.connect(db)
for row in con.execute("select name,counter_name,avg(counter_value) from pmc_events where name like '%walk_ragged%' group by name,counter_name"): print('waves=$w', row[0][:60], row[1], round(row[2]))
PY
done >> $OLDPWD/gpurun_out/r07l_ragged_waves.txt 2>&1
tail -12 $OLDPWD/gpurun_out/r07l_ragged_waves.txt
