#!/bin/bash
# tools/gpu_session.sh -- one gpurun call's worth of work (run on the GPU box from the repo root):
#   tools/gpu_session.sh <tag> [steps...]     steps: test bench ragged c5 eager
# Every step runs under its own timeout and writes to gpurun_out/<tag>_*.
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
for step in "$@"; do
	case $step in
	test)   timeout 1000 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log; tail -5 gpurun_out/${TAG}_pytest.log ;;
	testr3) timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -s --timeout 500 > gpurun_out/${TAG}_pytest_r3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_r3.log; tail -25 gpurun_out/${TAG}_pytest_r3.log ;;
	pkdebug) for a in ${PKDEBUG_CASES:-c1.npz:0:len40to0}; do a=$(echo $a | tr ':' ' '); FSM_HIP_DEBUG=2 timeout 200 python tests/tools/packed_debug.py $a > gpurun_out/${TAG}_pkdebug_$(echo $a | tr ' .' '__').log 2>&1; echo "pkdebug $a rc=$?"; grep -v "^  File\|^Extension" gpurun_out/${TAG}_pkdebug_$(echo $a | tr ' .' '__').log | tail -${PKDEBUG_TAIL:-14}; done ;;
	pkprof) export RAGGED_N=6000000 RAGGED_DISTS=${PK_DISTS:-short8-64} RAGGED_CASES=${PK_CASES:-c2:packed:4,c3:packed:4}
	        (cd /tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/${TAG}_pktrace -o t -- python $OLDPWD/tests/tools/ragged.py > $OLDPWD/gpurun_out/${TAG}_pktrace.txt 2>&1)
	        python tools/rocpd_top_kernels.py gpurun_out/${TAG}_pktrace >> gpurun_out/${TAG}_pktrace.txt 2>&1; rm -rf gpurun_out/${TAG}_pktrace; tail -12 gpurun_out/${TAG}_pktrace.txt
	        timeout 600 python tools/pmc_kernel.py --match packed --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" -- python tests/tools/ragged.py > gpurun_out/${TAG}_pkpmc.txt 2>&1; tail -70 gpurun_out/${TAG}_pkpmc.txt ;;
	pkbisect) export RAGGED_N=6000000 RAGGED_DISTS=short8-64 RAGGED_CASES=c2:packed:4
	        for dbg in ${PK_DBGS:-0 1 4 5}; do echo "PK_DEBUG=$dbg"; PK_DEBUG=$dbg timeout 200 python tests/tools/ragged.py 2>&1 | grep "mode= 4 waves= 0"; done > gpurun_out/${TAG}_pkbisect.txt 2>&1; cat gpurun_out/${TAG}_pkbisect.txt ;;
	testr3dbg) FSM_HIP_DEBUG=2 timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -s --timeout 500 -k length_distributions > gpurun_out/${TAG}_pytest_r3dbg.log 2>&1; grep -v "^  File\|^Extension" gpurun_out/${TAG}_pytest_r3dbg.log | tail -30 ;;
	pksweep) export RAGGED_DISTS=short8-64 RAGGED_CASES=${PK_CASES:-c2:packed:4,c3:packed:4}
	        SW=${PK_SWEEP:-6000000:16:0,6000000:12:9,6000000:16:8,6000000:8:10,24000000:16:0,24000000:12:9}
	        for cfg in $(echo $SW | tr ',' ' '); do
	            n=${cfg%%:*}; r=${cfg#*:}; w=${r%%:*}; x=${r#*:}
	            echo "== n=$n waves=$w rmax=$x"; RAGGED_N=$n PK_WAVES=$w PK_RMAX=$x timeout 300 python tests/tools/ragged.py 2>&1 | grep "mode= 4 waves= 0"
	        done > gpurun_out/${TAG}_pksweep.txt 2>&1; cat gpurun_out/${TAG}_pksweep.txt ;;
	testgen) timeout 1200 python -m pytest tests -m gpu -x -q --timeout 600 -k "ragged or packed or batch_sizes or golden_vectors or fuzz or arena or unaligned or error_contracts or large_batch or resume or eager_outputs_golden or endids or c_program or retest_style" > gpurun_out/${TAG}_pytest_gen.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gen.log; grep -v "^  File\|^Extension" gpurun_out/${TAG}_pytest_gen.log | tail -25 ;;
	c5ab)   timeout 500 python tests/tools/c5_probe.py --layout 7 --n 4000000 --variants "${C5_VARIANTS:-20=0;20=1;20=3;20=3,4=12;20=1}" > gpurun_out/${TAG}_c5ab.txt 2>&1; echo "c5ab rc=$?"; tail -12 gpurun_out/${TAG}_c5ab.txt
	        timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config5 or literal_set or aho_corasick" > gpurun_out/${TAG}_c5tests.log 2>&1; tail -3 gpurun_out/${TAG}_c5tests.log ;;
	c5mem)  timeout 900 python tools/pmc_kernel.py --match walk_direct --sets "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE;TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum;TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCC_BUSY_avr TCC_TAG_STALL_sum" --out gpurun_out/${TAG}_c5mem.json -- python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --subs none --knob 20=${C5_SF:-1} > gpurun_out/${TAG}_c5mem.txt 2>&1; tail -40 gpurun_out/${TAG}_c5mem.txt ;;
	c5pmc)  for sf in ${C5_SF:-0 1}; do timeout 600 python tools/pmc_kernel.py --match walk_direct --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT GRBM_GUI_ACTIVE" --out gpurun_out/${TAG}_c5pmc_sf$sf.json -- python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --subs none --knob 20=$sf > gpurun_out/${TAG}_c5pmc_sf$sf.txt 2>&1; tail -45 gpurun_out/${TAG}_c5pmc_sf$sf.txt; done ;;
	genpmc) export RAGGED_N=6000000 RAGGED_DISTS=${PK_DISTS:-short8-64} RAGGED_CASES=${PK_CASES:-c2:packed:2,c3:packed:2}
	        timeout 600 python tools/pmc_kernel.py --match walk_generic --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" -- python tests/tools/ragged.py > gpurun_out/${TAG}_genpmc.txt 2>&1; tail -62 gpurun_out/${TAG}_genpmc.txt ;;
	testr2) timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -x -q > gpurun_out/${TAG}_pytest_r2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_r2.log; tail -15 gpurun_out/${TAG}_pytest_r2.log ;;
	bench)  timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/${TAG}_bench.json ;;
	fullpar) timeout 900 python bench.py --steps 5 --warmup 2 --subs none --full-parity > gpurun_out/${TAG}_bench_fullparity.json 2> gpurun_out/${TAG}_bench_fullparity.err; echo "fullparity rc=$?"; python -c "import json,sys; r=json.loads(open('gpurun_out/${TAG}_bench_fullparity.json').read().strip().splitlines()[-1]); print(r['value'], r.get('full_parity'))" ;;
	benchnode) FSM_BENCH_NODE_FRONT=1 FSM_BENCH_NODE_REPLICAS=2 timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_benchnode.json 2> gpurun_out/${TAG}_benchnode.err; echo "benchnode rc=$?"; python -c "import json; r=json.loads(open('gpurun_out/${TAG}_benchnode.json').read().strip().splitlines()[-1]); print(r['value'], r.get('node_front'))" ;;
	ragged) timeout 400 python tests/tools/ragged.py > gpurun_out/${TAG}_ragged.txt 2>&1; echo "ragged rc=$?"; tail -30 gpurun_out/${TAG}_ragged.txt ;;
	c5)     timeout 400 python tests/tools/c5_probe.py --layout 7 --n 2000000 --variants "10=0,2=4;10=1,2=4;10=1,2=8;1=1" > gpurun_out/${TAG}_c5.txt 2>&1; echo "c5 rc=$?"; tail -12 gpurun_out/${TAG}_c5.txt ;;
	c3tab)  for kv in 6=1 6=9 6=1; do timeout 300 python bench.py --workload c3t --steps 10 --warmup 2 --no-cpu-baseline --subs none --no-full-parity --knob $kv > gpurun_out/${TAG}_c3tab_$kv.json 2> gpurun_out/${TAG}_c3tab_$kv.err; python -c "import json; r=json.loads(open('gpurun_out/${TAG}_c3tab_$kv.json').read().strip().splitlines()[-1]); print('c3t knob $kv', r['value'], r['unit'], r['roofline']['achieved'], r.get('parity'))"; done ;;
	eagerpmc) timeout 600 python tools/pmc_kernel.py --match walk_ldsdma --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" --out gpurun_out/${TAG}_eagerpmc.json -- python tests/tools/eager_probe.py --k 40 --layouts 0 --waves ${EAGER_WAVES:-0} > gpurun_out/${TAG}_eagerpmc.txt 2>&1; tail -70 gpurun_out/${TAG}_eagerpmc.txt ;;
	c3tpmc) timeout 600 python tools/pmc_kernel.py --match walk_direct --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" --out gpurun_out/${TAG}_c3tpmc.json -- python bench.py --workload c3t --steps 2 --warmup 1 --no-cpu-baseline --subs none --no-full-parity --knob ${C3T_KNOB:-6=1} > gpurun_out/${TAG}_c3tpmc.txt 2>&1; tail -40 gpurun_out/${TAG}_c3tpmc.txt ;;
	c3tsweep) timeout 400 python tests/tools/sweep.py --set c3t --workloads c3t --layouts ${C3T_LAYOUTS:-5} > gpurun_out/${TAG}_c3tsweep.txt 2>&1; echo "sweep rc=$?"; grep -v '^#' gpurun_out/${TAG}_c3tsweep.txt | tail -30 ;;
	eager)  timeout 300 python tests/tools/eager_probe.py --layouts 0 --waves ${EAGER_WAVES:-0,12} > gpurun_out/${TAG}_eager.txt 2>&1; echo "eager rc=$?"; tail -12 gpurun_out/${TAG}_eager.txt ;;
	sweep)  timeout 400 python tests/tools/sweep.py --set r2 --workloads c3,c2 > gpurun_out/${TAG}_sweep.txt 2>&1; echo "sweep rc=$?"; grep -v '^#' gpurun_out/${TAG}_sweep.txt | tail -80 ;;
	prof)   for wl in c3 c2 c5; do
	            L2=""; [ $wl = c5 ] && L2="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
	            L2_PMC="$L2" timeout 500 bash tools/profile.sh $wl $PWD/gpurun_out/${TAG}_prof_$wl > gpurun_out/${TAG}_prof_$wl.log 2>&1
	            python tools/rocpd_summary.py gpurun_out/${TAG}_prof_$wl $wl gpurun_out/${TAG}_$wl >> gpurun_out/${TAG}_prof_$wl.log 2>&1
	            tail -2 gpurun_out/${TAG}_prof_$wl.log
	            rm -rf gpurun_out/${TAG}_prof_$wl/*/*.db gpurun_out/${TAG}_prof_$wl/*/*/*.db 2>/dev/null
	        done ;;
	profn)  for wl in ${PROF_WLS:-c3 c3t c2 c5 c3_short c2_short c3_ragged c2_ragged c5_short c5_ragged c3_eager40}; do
	            L2=""; [ $wl = c5 ] && L2="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
	            FSM_BENCH_LINES_FORMS=off64 L2_PMC="$L2" timeout 600 bash tools/profile.sh $wl $PWD/gpurun_out/${TAG}_prof_$wl > gpurun_out/${TAG}_prof_$wl.log 2>&1
	            python tools/rocpd_summary.py gpurun_out/${TAG}_prof_$wl $wl gpurun_out/${TAG}_$wl >> gpurun_out/${TAG}_prof_$wl.log 2>&1
	            tail -2 gpurun_out/${TAG}_prof_$wl.log
	            rm -rf gpurun_out/${TAG}_prof_$wl/*/*.db gpurun_out/${TAG}_prof_$wl/*/*/*.db 2>/dev/null
	        done ;;
	profls) timeout 500 bash tools/profile.sh c3 $PWD/gpurun_out/${TAG}_prof_c3_loadskip --knob 6=3 > gpurun_out/${TAG}_prof_c3_loadskip.log 2>&1
	        python tools/rocpd_summary.py gpurun_out/${TAG}_prof_c3_loadskip c3_loadskip gpurun_out/${TAG}_c3_loadskip >> gpurun_out/${TAG}_prof_c3_loadskip.log 2>&1
	        tail -2 gpurun_out/${TAG}_prof_c3_loadskip.log; rm -rf gpurun_out/${TAG}_prof_c3_loadskip/*/*.db gpurun_out/${TAG}_prof_c3_loadskip/*/*/*.db 2>/dev/null ;;
	calib)  cd /tmp; rocprofv3 --pmc FETCH_SIZE -d $OLDPWD/gpurun_out/${TAG}_calib/fetch -o f -- python $OLDPWD/tools/fetch_calib.py > $OLDPWD/gpurun_out/${TAG}_calib_run.txt 2>&1
	        rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $OLDPWD/gpurun_out/${TAG}_calib/req -o r -- python $OLDPWD/tools/fetch_calib.py >> $OLDPWD/gpurun_out/${TAG}_calib_run.txt 2>&1
	        cd $OLDPWD; python tools/fetch_calib.py --read gpurun_out/${TAG}_calib/fetch gpurun_out/${TAG}_calib/req > gpurun_out/${TAG}_fetch_calib.json 2>&1; cat gpurun_out/${TAG}_fetch_calib.json | tail -12
	        rm -rf gpurun_out/${TAG}_calib ;;
	smoke)  timeout 300 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -6 gpurun_out/${TAG}_smoke.log ;;
	node)   timeout 300 python -m pytest tests/test_gpu_round2.py -m gpu -q -k "node or retest" > gpurun_out/${TAG}_node.log 2>&1; tail -5 gpurun_out/${TAG}_node.log ;;
	r4lazy) timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q --timeout 600 > gpurun_out/${TAG}_pytest_r4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_r4.log; grep -v "^  File\|^Extension" gpurun_out/${TAG}_pytest_r4.log | tail -25 ;;
	c5lazy) timeout 600 python tests/tools/c5_probe.py --layout 7 --n ${C5_N:-4000000} --variants "${C5_VARIANTS:-20=3;20=1;20=3}" > gpurun_out/${TAG}_c5lazy.txt 2>&1; echo "c5lazy rc=$?"; tail -12 gpurun_out/${TAG}_c5lazy.txt ;;
	c5pmcl) for v in ${C5_PMC_VARIANTS:-20=3,3=2 20=3,3=3}; do timeout 600 python tools/pmc_kernel.py --match walk_lazy --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE;TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" --out gpurun_out/${TAG}_c5pmcl_$v.json -- python tests/tools/c5_probe.py --layout 7 --n 2000000 --reps 2 --variants "$v" > gpurun_out/${TAG}_c5pmcl_$v.txt 2>&1; tail -48 gpurun_out/${TAG}_c5pmcl_$v.txt; done ;;
	counters) (cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "icache|SQC_|IFETCH|INST_CACHE" | head -60) > gpurun_out/${TAG}_counters.txt 2>&1; head -50 gpurun_out/${TAG}_counters.txt ;;
	genic) export RAGGED_N=${RAGGED_N:-6000000} RAGGED_DISTS=${PK_DISTS:-short8-64} RAGGED_CASES=${PK_CASES:-c2:packed:2,c3:packed:2}
	        timeout 600 python tools/pmc_kernel.py --match walk_generic --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_IFETCH SQ_INSTS_SMEM;SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ GRBM_GUI_ACTIVE;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQC_ICACHE_BUSY_CYCLES" --out gpurun_out/${TAG}_genic.json -- python tests/tools/ragged.py > gpurun_out/${TAG}_genic.txt 2>&1; tail -90 gpurun_out/${TAG}_genic.txt ;;
	lds2)   timeout 500 python tests/tools/lds2_probe.py > gpurun_out/${TAG}_lds2.txt 2>&1; echo "lds2 rc=$?"; tail -24 gpurun_out/${TAG}_lds2.txt ;;
	benchfull) timeout 1500 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/${TAG}_bench_default.json; tail -3 gpurun_out/${TAG}_bench_default.err ;;
	lines32ab) export RAGGED_N=${RAGGED_N:-24000000} RAGGED_DISTS=${PK_DISTS:-short8-64,short8-16} RAGGED_CASES=${PK_CASES:-c2:packed:-1,c3:packed:-1,c2:off32:-1,c3:off32:-1,c2:lengths:-1,c3:lengths:-1}
	        # early knob: -1 the shipped choice, 33 walk_generic's own body, 257 no second tile in flight, 4097 every lane asks for every chunk, 65 first-chunk skip tests kept
	        RAGGED_EARLY=${LINES32_EARLY:--1,33,257,4097} timeout 500 python tests/tools/ragged.py 2>&1 | grep "front=" > gpurun_out/${TAG}_lines32_ab.txt; cut -c1-160 gpurun_out/${TAG}_lines32_ab.txt ;;
	lines32x) export RAGGED_N=12000000 RAGGED_DISTS=short32-128,mid64-128,mid64-192,mid64-256,mid128-256 RAGGED_CASES=c2:packed:-1,c2:packed:2,c2:packed:3,c3:packed:-1,c3:packed:2,c3:packed:3
	        timeout 500 python tests/tools/ragged.py 2>&1 | grep "front=" > gpurun_out/${TAG}_lines32_crossover.txt; cut -c1-160 gpurun_out/${TAG}_lines32_crossover.txt ;;
	linesstress) LAYOUTS=${LAYOUTS:-0,1,2,3,4,5,6,8,9} MIXES=${MIXES:-0-200,8-64,8-16} MODES=${MODES:--1,2} REPS=${REPS:-100} timeout 1300 python tests/tools/lines_stress.py > gpurun_out/${TAG}_lines_stress.txt 2>&1; echo "stress rc=$?"; tail -2 gpurun_out/${TAG}_lines_stress.txt ;;
	c5tests) timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config5 or literal_set or aho_corasick" > gpurun_out/${TAG}_c5tests.log 2>&1; tail -3 gpurun_out/${TAG}_c5tests.log ;;
	esac
done
