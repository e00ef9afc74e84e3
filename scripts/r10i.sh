#!/bin/bash
# round 4, call i: what the wave-per-row kernel's time is made of (developer build, WRONG results on purpose): no index stores / no value stores / no adds
export SPRS_HIP_LIBRARY=$PWD/sprs_amd/libsprs_hip_dev.so
mkdir -p gpurun_out/r10i
for d in 0 2 4 8 6 14; do
  echo "-- spgemm_debug=$d"
  ( cd /tmp && rm -rf /tmp/st && SPGEMM_DEBUG=$d SPGEMM_HEAVY=524288 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 8 0 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "rows_kernel" | cut -c1-170
done | tee gpurun_out/r10i/mid_kernel_store_experiments.txt
