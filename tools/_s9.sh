mkdir -p gpurun_out
export RAGGED_N=6000000 RAGGED_DISTS=short8-64,short8-16,uniform0-1024 RAGGED_CASES=c3:packed:2,c3:packed:-1
for lay in 0 3 5; do echo "== layout flag $lay"; RAGGED_LAYOUT=$lay timeout 200 python tests/tools/ragged.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r07i_c3_lines_by_layout.txt 2>&1
cat gpurun_out/r07i_c3_lines_by_layout.txt
