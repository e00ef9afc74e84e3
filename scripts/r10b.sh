#!/bin/bash
# round 4, call b: lean wave-per-row kernel (mark + max-scan owners, lane-order adds): parity gate, timing, kernel stats, A/B, phases
bash scripts/gpu_session.sh r10b "gate:test_spgemm_gpu" "spgemm_ab:base|SPGEMM_LANE_ORDER=2|SPGEMM_MID=262144|SPGEMM_MID=1048576" spgemm_stats
SPRS_HIP_LIBRARY=$PWD/sprs_amd/libsprs_hip_dev.so SPGEMM_PROF=1 timeout 600 python tests/spgemm_bench.py 1000000 8 8 1 2>&1 | grep -E "spgemm_prof.*mid|seconds" | cut -c1-400 | tee gpurun_out/r10b/spgemm_prof.txt
