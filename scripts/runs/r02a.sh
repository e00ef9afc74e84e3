#!/bin/bash
# round 2, GPU call a: banded SpMV plan — parity on the device, option sweep on R-MAT 10M, kernel trace, PMC
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02a
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
echo "== pytest band + spmv"
timeout 900 python -m pytest tests/test_spmv_band_gpu.py tests/test_spmv_gpu.py -m gpu -x -q 2>&1 | tail -8
echo "== sweep"
timeout 900 python scripts/spmv_sweep.py --oracle \
  "band24:" "old_sliced:spmv_band=2" \
  "band8:spmv_band_hot=8" "band16:spmv_band_hot=16" "band32:spmv_band_hot=32" "band48:spmv_band_hot=48" "band64:spmv_band_hot=64" "band96:spmv_band_hot=96" \
  "b24_ph2:spmv_band_phases=2" "b24_ph3:spmv_band_phases=3" "b48_ph2:spmv_band_hot=48,spmv_band_phases=2" \
  "b24_g1:spmv_band_group=1" "b24_g2:spmv_band_group=2" "b24_g8:spmv_band_group=8" "b24_g16:spmv_band_group=16" \
  "b24_s64:spmv_xcs_split=64" "b24_s16:spmv_xcs_split=16" "b48_s64:spmv_band_hot=48,spmv_xcs_split=64" \
  2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.jsonl
echo "== kernel trace, band24"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmv_sweep.py --steps 10 "band24:" > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|^#|sprs_hip" | cut -c1-200 | tee $OUT/kernel_stats_band24.txt
echo "== kernel trace, band48 phases 2"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmv_sweep.py --steps 10 "b48_ph2:spmv_band_hot=48,spmv_band_phases=2" > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|^#|sprs_hip" | cut -c1-200 | tee $OUT/kernel_stats_b48_ph2.txt
echo "== PMC, band24"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  ( cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/scripts/spmv_sweep.py --steps 3 --warmup 1 "band24:" > /dev/null 2> $OUT/pmc$i.err )
  f=$(find /tmp/pmc_$i -name "*.db" | head -1)
  echo "-- group $i: $grp"
  if [ -n "$f" ]; then python3 scripts/rocprof_summary.py "$f" sprs_hip | sed -n '/PMC counters/,$p' | cut -c1-200; else tail -3 $OUT/pmc$i.err; fi
done 2>&1 | tee $OUT/pmc_band24.txt
} 2>&1 | tee $OUT/log.txt
