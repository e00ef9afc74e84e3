#!/bin/bash
# round 4, call c: packed B (u32 columns for the counting walks, {column, value} records for the value walks), chunks of 8 wave instructions
bash scripts/gpu_session.sh r10c "gate:test_spgemm_gpu" "spgemm_ab:base|SPGEMM_MID=262144" spgemm_stats spgemm_traffic1
SPRS_HIP_LIBRARY=$PWD/sprs_amd/libsprs_hip_dev.so SPGEMM_PROF=1 timeout 600 python tests/spgemm_bench.py 1000000 8 8 1 2>&1 | grep -E "spgemm_prof.*(mid|class|large)|seconds" | cut -c1-400 | tee gpurun_out/r10c/spgemm_prof.txt
