#!/bin/bash
# round 4, call a: LDS lane-order probe + config-5 baseline on this box + phase profile of the current kernels (developer build)
OUT=gpurun_out/r10a; mkdir -p $OUT
./scripts/probes/lds_add_order.out 200000 > $OUT/lds_add_order.jsonl 2>&1
cat $OUT/lds_add_order.jsonl
timeout 600 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep -E "seconds" | tee $OUT/spgemm_base.jsonl
SPRS_HIP_LIBRARY=$PWD/sprs_amd/libsprs_hip_dev.so SPGEMM_PROF=1 timeout 600 python tests/spgemm_bench.py 1000000 8 8 1 2>&1 | grep -E "spgemm_prof|seconds" | tee $OUT/spgemm_prof.txt
