#!/bin/bash
# SpMM entry-stream kernel: parity gate, then the bound probe (L2-resident rhs / uniform / R-MAT) for both kernels, then kernel stats
mkdir -p gpurun_out/r11d
timeout 600 python -m pytest tests/test_spmm_gpu.py -m gpu -x -q 2>&1 | tail -4
if [ "${PIPESTATUS[0]}" != 0 ]; then echo "gate failed"; exit 1; fi
timeout 600 python scripts/spmm_bound_probe.py 8 16 32 2>&1 | grep -v amdgpu | tee gpurun_out/r11d/spmm_stream.jsonl
SPRS_OPTS="spmm_stream=0" timeout 600 python scripts/spmm_bound_probe.py 16 2>&1 | grep -v amdgpu | tee gpurun_out/r11d/spmm_chunks.jsonl
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/st && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmm_bench.py 10000000 32 8 16 > gpurun_out_spmm.json 2>/dev/null; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|^#|spmm|tile_rows" | cut -c1-200 | tee gpurun_out/r11d/spmm_kernel_stats.txt
