#!/bin/bash
# round 4, call l: phase profile of the workgroup kernel on the current build; traffic
export SPRS_HIP_LIBRARY=$PWD/sprs_amd/libsprs_hip_dev.so
mkdir -p gpurun_out/r10l
for e in "SPGEMM_DEBUG=0" "SPGEMM_DEBUG=2" "SPGEMM_ORDERED=0"; do
echo "-- $e"
env $e SPGEMM_PROF=1 timeout 600 python tests/spgemm_bench.py 1000000 8 8 0 2>&1 | grep -E "spgemm_prof" | grep -v "top\|class [034]" | cut -c1-330 | head -8
done | tee gpurun_out/r10l/spgemm_prof.txt
unset SPRS_HIP_LIBRARY
bash scripts/gpu_session.sh r10l spgemm_traffic1
