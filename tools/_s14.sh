mkdir -p gpurun_out
timeout 1300 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r07m_pytest_all.log 2>&1; tail -6 gpurun_out/r07m_pytest_all.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r07m_smoke.log 2>&1; echo "smoke rc=$?"; grep -v amdgpu.ids gpurun_out/r07m_smoke.log | tail -6
RAGGED_N=6000000 RAGGED_DISTS=uniform0-1024 RAGGED_CASES=c2:packed:3,c2:stride+len:3,c3:packed:3,c3:stride+len:3 timeout 200 python tests/tools/ragged.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r07m_ragged_packed_vs_aligned_rows.txt; cat gpurun_out/r07m_ragged_packed_vs_aligned_rows.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r07m_bench_default.json 2> gpurun_out/r07m_bench_default.err; echo "bench rc=$?"; wc -c gpurun_out/r07m_bench_default.json; cp bench_detail.json gpurun_out/r07m_bench_detail.json; tail -c 1200 gpurun_out/r07m_bench_default.json
