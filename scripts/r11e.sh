#!/bin/bash
# SpMM entry-stream kernel, second cut (vector LDS reads in the adds, empty rows by their own kernel, fix-up by lane groups): gate, bench with and without the profiler, probe
mkdir -p gpurun_out/r11e
timeout 600 python -m pytest tests/test_spmm_gpu.py -m gpu -x -q 2>&1 | tail -3
if [ "${PIPESTATUS[0]}" != 0 ]; then echo "gate failed"; exit 1; fi
timeout 600 python scripts/spmm_bench.py 10000000 32 8 16 32 64 2>&1 | grep -v amdgpu | tee gpurun_out/r11e/spmm_bench.jsonl
timeout 600 python scripts/spmm_bound_probe.py 8 16 2>&1 | grep -v amdgpu | tee gpurun_out/r11e/spmm_stream.jsonl
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/st && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmm_bench.py 10000000 32 8 16 > $GRAFT_REPO_ROOT/gpurun_out/r11e/spmm_bench_under_rocprof.jsonl 2>/dev/null; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|^#|spmm|tile_rows" | cut -c1-200 | tee gpurun_out/r11e/spmm_kernel_stats.txt
cat gpurun_out/r11e/spmm_bench_under_rocprof.jsonl
