mkdir -p gpurun_out
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -o -E "TCC_EA0?_[A-Z0-9_]*|TCC_[A-Z_]*MALL[A-Z_0-9]*|TCC_[A-Z_]*DRAM[A-Z_0-9]*|TCC_BUBBLE[A-Z_]*|TCC_[A-Z_]*128B[A-Z_]*|TCC_[A-Z_]*64B[A-Z_]*|TCC_[A-Z_]*32B[A-Z_]*" | sort -u) > gpurun_out/r07h_tcc_counters.txt 2>&1
wc -l gpurun_out/r07h_tcc_counters.txt; head -80 gpurun_out/r07h_tcc_counters.txt | tr '\n' ' '
timeout 200 python bench.py --workload c3_eager40 --steps 5 --warmup 1 > gpurun_out/r07h_eager40.json 2> gpurun_out/r07h_eager40.err; echo "eager40 rc=$?"; tail -c 1200 gpurun_out/r07h_eager40.json; tail -2 gpurun_out/r07h_eager40.err | cut -c 1-400
timeout 700 python bench.py --steps 20 --warmup 5 > gpurun_out/r07h_bench_default.json 2> gpurun_out/r07h_bench_default.err; echo "bench rc=$?"; wc -c gpurun_out/r07h_bench_default.json; cp bench_detail.json gpurun_out/r07h_bench_detail.json; tail -c 2500 gpurun_out/r07h_bench_default.json
