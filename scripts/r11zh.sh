#!/bin/bash
# SpGEMM config 5, natural labels: the wave kernel's rows drawn in a scattered order (developer build, spgemm_debug = 16) against row order
mkdir -p gpurun_out/r11zh
export SPRS_HIP_LIBRARY=$GRAFT_REPO_ROOT/sprs_amd/libsprs_hip_dev.so
export TMPDIR=/tmp
for d in 0 16; do
  echo "== SPGEMM_DEBUG=$d" | tee -a gpurun_out/r11zh/spgemm_scattered_row_order.txt
  SPGEMM_DEBUG=$d timeout 300 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep -E "seconds" | cut -c1-200 | tee -a gpurun_out/r11zh/spgemm_scattered_row_order.txt
  ( cd /tmp && rm -rf /tmp/st && SPGEMM_DEBUG=$d timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 8 1 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "mid_rows_kernel|large_rows" | cut -c1-200 | tee -a gpurun_out/r11zh/spgemm_scattered_row_order.txt
done
