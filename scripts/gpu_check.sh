#!/bin/bash
# One gpurun call: GPU parity tests, bench lines (default + A/B), rocprof kernel stats.
# Usage on the GPU box (from the repo root): bash scripts/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo" ; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
echo "== bench default (rmat10m)"
timeout 900 python bench.py --steps 30 --warmup 3 2>$OUT/bench_default.err | tee $OUT/bench_default.json
tail -3 $OUT/bench_default.err
for variant in "--kernel 2" "--tile 2048" "--nt 0" "--idx-bytes 4"; do
  name=$(echo $variant | tr -d ' -')
  echo "== bench rmat10m $variant"
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $variant 2>$OUT/bench_$name.err | tee $OUT/bench_$name.json
done
echo "== bench laplace4096 / rmat1m"
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --workload laplace4096 2>>$OUT/bench_misc.err | tee $OUT/bench_laplace4096.json
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --workload rmat1m 2>>$OUT/bench_misc.err | tee $OUT/bench_rmat1m.json
echo "== rocprofv3 kernel stats"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o spmv -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2>$GRAFT_REPO_ROOT/$OUT/prof.err )
cat $OUT/prof_bench.json
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
