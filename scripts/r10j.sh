#!/bin/bash
# round 4, call j: phase profile of the wave-per-row kernel after the unconditional loads
export SPRS_HIP_LIBRARY=$PWD/sprs_amd/libsprs_hip_dev.so
mkdir -p gpurun_out/r10j
for e in "SPGEMM_DEBUG=0" "SPGEMM_DEBUG=14" "SPGEMM_MID_KEEP=4"; do
echo "-- $e"
env $e SPGEMM_HEAVY=524288 SPGEMM_PROF=1 timeout 600 python tests/spgemm_bench.py 1000000 8 8 0 2>&1 | grep -E "spgemm_prof.*mid" | cut -c1-330 | head -2
done | tee gpurun_out/r10j/spgemm_prof.txt
