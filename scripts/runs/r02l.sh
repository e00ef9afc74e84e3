#!/bin/bash
# round 2, GPU call l: blocks per workgroup of the tiled short-rows launch; overlap on/off
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02l
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
echo "== sweep"
timeout 900 python scripts/spmv_sweep.py --steps 30 \
  "sg4:" "sg1:spmv_band_short_group=1" "sg2:spmv_band_short_group=2" "sg8:spmv_band_short_group=8" "short_gather:spmv_band_short=2" \
  "sg2_ov:spmv_band_short_group=2,spmv_band_overlap=1" "short_gather_ov:spmv_band_short=2,spmv_band_overlap=1" \
  "sg4_again:" "sg2_again:spmv_band_short_group=2" "short_gather_again:spmv_band_short=2" "short_gather_ov_again:spmv_band_short=2,spmv_band_overlap=1" \
  2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee $OUT/sweep.jsonl
} 2>&1 | tee $OUT/log.txt
