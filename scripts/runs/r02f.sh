#!/bin/bash
# round 2, GPU call f: cold + short rows as register wave tiles (high occupancy), batched reduce tables
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02f
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
echo "== pytest band"
timeout 900 python -m pytest tests/test_spmv_band_gpu.py -m gpu -x -q 2>&1 | tail -4
echo "== sweep"
timeout 900 python scripts/spmv_sweep.py --oracle \
  "b64_s64:spmv_band_hot=64,spmv_xcs_split=64" "b64_s64_t512:spmv_band_hot=64,spmv_xcs_split=64,spmv_band_hot_threads=512" \
  "b96_s64:spmv_band_hot=96,spmv_xcs_split=64" "b96_s32:spmv_band_hot=96" "b64_s32:spmv_band_hot=64" "b48_s32:spmv_band_hot=48" "b24_s32:" \
  "b96_s48:spmv_band_hot=96,spmv_xcs_split=48" "b64_s48:spmv_band_hot=64,spmv_xcs_split=48" "b64_s96:spmv_band_hot=64,spmv_xcs_split=96" \
  "b96_s16:spmv_band_hot=96,spmv_xcs_split=16" "b96_s24:spmv_band_hot=96,spmv_xcs_split=24" \
  "b64_s64_g2:spmv_band_hot=64,spmv_xcs_split=64,spmv_band_group=2" "b64_s64_g8:spmv_band_hot=64,spmv_xcs_split=64,spmv_band_group=8" \
  2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.jsonl
for cfg in "b64_s64:spmv_band_hot=64,spmv_xcs_split=64,spmv_band_split_launch=1" "b96_s32:spmv_band_hot=96,spmv_band_split_launch=1"; do
echo "== kernel trace, $cfg"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmv_sweep.py --steps 10 "$cfg" > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_seq.py $(find /tmp/st -name "*.db" | head -1) band_ ) 2>&1 | cut -c1-200
done | tee $OUT/kernel_seq.txt
} 2>&1 | tee $OUT/log.txt
