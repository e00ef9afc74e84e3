#!/bin/bash
# round 2, GPU call g: more hot slices (up to 384), reduce with scalar broadcasts
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02g
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
echo "== sweep"
SPRS_HIP_DEBUG=1 timeout 900 python scripts/spmv_sweep.py --oracle \
  "b96_s48:spmv_band_hot=96,spmv_xcs_split=48" "b128_s48:spmv_band_hot=128,spmv_xcs_split=48" "b192_s48:spmv_band_hot=192,spmv_xcs_split=48" "b256_s48:spmv_band_hot=256,spmv_xcs_split=48" "b384_s48:spmv_band_hot=384,spmv_xcs_split=48" \
  "b128_s32:spmv_band_hot=128" "b192_s32:spmv_band_hot=192" "b256_s32:spmv_band_hot=256" \
  "b128_s64:spmv_band_hot=128,spmv_xcs_split=64" "b192_s64:spmv_band_hot=192,spmv_xcs_split=64" "b256_s64:spmv_band_hot=256,spmv_xcs_split=64" \
  "b192_s48_g16:spmv_band_hot=192,spmv_xcs_split=48,spmv_band_group=16" "b192_s48_g4:spmv_band_hot=192,spmv_xcs_split=48,spmv_band_group=4" \
  2>&1 | grep -v amdgpu.ids | cut -c1-1500 | tee $OUT/sweep.jsonl
for cfg in "b192_s48:spmv_band_hot=192,spmv_xcs_split=48,spmv_band_split_launch=1"; do
echo "== kernel trace, $cfg"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmv_sweep.py --steps 10 "$cfg" > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_seq.py $(find /tmp/st -name "*.db" | head -1) band_ ) 2>&1 | cut -c1-200
done | tee $OUT/kernel_seq.txt
} 2>&1 | tee $OUT/log.txt
