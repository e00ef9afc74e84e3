#!/bin/bash
# the slow mode of the k = 16 SpMM call (spmm_bench with k = 16 alone): do the non-temporal hints matter there?  developer build
mkdir -p gpurun_out/r11v
export SPRS_HIP_LIBRARY=$GRAFT_REPO_ROOT/sprs_amd/libsprs_hip_dev.so
for o in "" "spmm_debug=1" "spmm_debug=2" "spmm_debug=3" "spmm_debug=4" "spmm_debug=7"; do
  echo "-- ${o:-defaults}" | tee -a gpurun_out/r11v/spmm_nt_ab.jsonl
  SPRS_OPTS="$o" timeout 200 python scripts/spmm_bench.py 10000000 32 16 2>&1 | grep -v amdgpu | tee -a gpurun_out/r11v/spmm_nt_ab.jsonl
done
echo "-- fast history (8 then 16), defaults and plain" | tee -a gpurun_out/r11v/spmm_nt_ab.jsonl
timeout 200 python scripts/spmm_bench.py 10000000 32 8 16 2>&1 | grep -v amdgpu | tee -a gpurun_out/r11v/spmm_nt_ab.jsonl
SPRS_OPTS="spmm_debug=3" timeout 200 python scripts/spmm_bench.py 10000000 32 8 16 2>&1 | grep -v amdgpu | tee -a gpurun_out/r11v/spmm_nt_ab.jsonl
