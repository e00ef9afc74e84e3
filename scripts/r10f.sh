#!/bin/bash
# round 4, call f: where the workgroup kernel's time goes with lane-order adds; A/B of its geometry
mkdir -p gpurun_out/r10f
SPRS_HIP_LIBRARY=$PWD/sprs_amd/libsprs_hip_dev.so SPGEMM_PROF=1 timeout 600 python tests/spgemm_bench.py 1000000 8 8 1 2>&1 | grep -E "spgemm_prof|seconds" | cut -c1-300 | head -24 | tee gpurun_out/r10f/spgemm_prof.txt
bash scripts/gpu_session.sh r10f "spgemm_ab:SPGEMM_HEAVY=262144|SPGEMM_HEAVY=524288|SPGEMM_HEAVY=262144 SPGEMM_OCCUPANCY=2|SPGEMM_HEAVY=262144 SPGEMM_WINLOG=16|SPGEMM_HEAVY=262144 SPGEMM_WINLOG=18|SPGEMM_HEAVY=262144 SPGEMM_RETAIN=0|SPGEMM_HEAVY=262144 SPGEMM_OVERLAP=1|SPGEMM_HEAVY=262144 SPGEMM_MINWIN=14"
