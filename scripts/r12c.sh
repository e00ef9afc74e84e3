#!/bin/bash
# round 4, the last session on the sources as committed: GPU suite + smoke, the driver's bench command, kernel stats of it, traffic PMC of the SpMV and of one config-5 product, whole-product parity, SpMM bench
bash scripts/gpu_session.sh r12c tests bench stats pmc spgemm_traffic1 spgemm_stats spgemm_parity
timeout 300 python scripts/spmm_bench.py 10000000 32 8 12 16 24 32 48 64 2>&1 | grep -v amdgpu | tee gpurun_out/r12c/spmm_bench.jsonl
( cd /tmp && rm -rf /tmp/st && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmm_bench.py 10000000 32 8 16 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|^#|spmm|tile_rows" | cut -c1-200 | tee gpurun_out/r12c/spmm_kernel_stats.txt
