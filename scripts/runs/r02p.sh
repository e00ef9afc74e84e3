#!/bin/bash
# round 2, GPU call p: SpMV checkpoint — whole GPU suite, smoke, bench lines (configs 4, 2 warm + cold, 3), trace, PMC
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02p
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -3
echo "== bench default (config 4)"
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tee $OUT/bench_default.json
echo "== bench again"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_default2.json
echo "== bench rmat1m warm / cold, laplace4096"
timeout 900 python bench.py --workload rmat1m --steps 50 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_rmat1m_warm.json
timeout 900 python bench.py --workload rmat1m --steps 50 --no-cpu-baseline --cold-cache 2>/dev/null | tee $OUT/bench_rmat1m_cold.json
timeout 900 python bench.py --workload rmat1m --steps 50 --no-cpu-baseline --cold-cache --band 1 2>/dev/null | tee $OUT/bench_rmat1m_cold_band.json
timeout 900 python bench.py --workload laplace4096 --steps 50 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_laplace4096.json
echo "== kernel trace of the default bench command"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats_bench.json 2>/dev/null; f=$(find /tmp/st -name "*.db" | head -1); python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $f sprs_hip | grep -E "^kernel|^#|band_" | cut -c1-190; python3 $GRAFT_REPO_ROOT/scripts/rocprof_seq.py $f band_ | cut -c1-200 ) 2>&1 | tee $OUT/kernel_stats.txt
echo "== PMC"
PMC_GROUPS=3 bash scripts/gpu_pmc.sh r02p/pmc > /dev/null 2>&1
grep -E "csrc_sha16|band_" gpurun_out/r02p/pmc/pmc_summary.txt | cut -c1-200
echo "== SpGEMM config 5 as it stands"
timeout 300 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep -E "seconds|nnz" | head -5
} 2>&1 | tee $OUT/log.txt
