#!/bin/bash
# One gpurun call that reproduces what the driver does at round end + the committed evidence:
# GPU parity tests, smoke(), the default bench line, rocprofv3 kernel stats of the same command.
# Usage on the GPU box (from the repo root): bash scripts/gpu_check.sh [tag]
TAG=${1:-check}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -14
echo "== smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -4
echo "== bench default"
timeout 900 python bench.py 2>/dev/null | tee $OUT/bench_default.json
echo "== rocprofv3 --kernel-trace --stats of the default bench command"
( cd /tmp && rm -rf /tmp/st && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/stats_bench.json 2>/dev/null; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|^#|sprs_hip" | cut -c1-190 | tee $OUT/kernel_stats.txt
} 2>&1 | tee $OUT/log.txt
