mkdir -p gpurun_out
export FSM_BENCH_LINES_FORMS=off64
for wl in c5_ragged c5_short; do
  for kn in 22=1 22=0; do
    timeout 300 python bench.py --workload $wl --n 8000000 --steps 5 --warmup 1 --no-cpu-baseline --knob $kn 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl knob $kn', r['value'], 'GB/s', r['ms_per_step'], 'ms', r['roofline']['kernel'])"
  done
done > gpurun_out/r07d_c5_lines_ab.txt 2>&1
cat gpurun_out/r07d_c5_lines_ab.txt
timeout 600 python tools/pmc_kernel.py --match walk_lazy --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT" --out gpurun_out/r07d_c5_ragged_pmc.json -- python bench.py --workload c5_ragged --n 8000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r07d_c5_ragged_pmc.txt 2>&1
tail -45 gpurun_out/r07d_c5_ragged_pmc.txt
