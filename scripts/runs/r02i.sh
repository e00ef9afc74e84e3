#!/bin/bash
# round 2, GPU call i: whole GPU suite + smoke + default bench line + sweep around the defaults + trace + PMC (3 groups)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02i
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -16
echo "== smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -4
echo "== bench default"
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tee $OUT/bench_default.json
echo "== sweep"
timeout 900 python scripts/spmv_sweep.py \
  "default:" "b96:spmv_band_hot=96" "b192:spmv_band_hot=192" "g8:spmv_band_group=8" "g32:spmv_band_group=32" "s24:spmv_xcs_split=24" "s48:spmv_xcs_split=48" "t512:spmv_band_hot_threads=512" \
  2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee $OUT/sweep.jsonl
echo "== kernel trace of the default bench command"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats_bench.json 2>/dev/null; f=$(find /tmp/st -name "*.db" | head -1); python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $f sprs_hip | grep -E "^kernel|^#|band_" | cut -c1-190; python3 $GRAFT_REPO_ROOT/scripts/rocprof_seq.py $f band_ | cut -c1-200 ) 2>&1 | tee $OUT/kernel_stats.txt
echo "== PMC"
PMC_GROUPS=3 bash scripts/gpu_pmc.sh r02i/pmc > /dev/null 2>&1
grep -E "csrc_sha16|band_" gpurun_out/r02i/pmc/pmc_summary.txt | cut -c1-200
} 2>&1 | tee $OUT/log.txt
