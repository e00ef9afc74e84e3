#!/bin/bash
# SpMM entry-stream kernel, empty rows folded in: gate, bench, kernel stats, L2 / fabric counters per kernel
mkdir -p gpurun_out/r11f
timeout 600 python -m pytest tests/test_spmm_gpu.py -m gpu -x -q 2>&1 | tail -3
if [ "${PIPESTATUS[0]}" != 0 ]; then echo "gate failed"; exit 1; fi
timeout 600 python scripts/spmm_bench.py 10000000 32 8 16 32 2>&1 | grep -v amdgpu | tee gpurun_out/r11f/spmm_bench.jsonl
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/st && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmm_bench.py 10000000 32 8 16 32 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|^#|spmm|tile_rows" | cut -c1-200 | tee gpurun_out/r11f/spmm_kernel_stats.txt
( cd /tmp && rm -rf /tmp/pm && timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --kernel-include-regex "spmm" -d /tmp/pm -o pmc -- python $GRAFT_REPO_ROOT/scripts/spmm_bench.py 10000000 32 8 16 32 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/pm -name "*.db" | head -1) sprs_hip | sed -n '/PMC counters/,$p' ) 2>&1 | cut -c1-220 | tee gpurun_out/r11f/spmm_pmc.txt
