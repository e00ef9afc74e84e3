#!/bin/bash
# round 2, GPU call h: gather-bound kernels on a second stream beside the HBM-bound hot kernel; reduce with 32 partials in flight
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02h
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
echo "== pytest band"
timeout 900 python -m pytest tests/test_spmv_band_gpu.py -m gpu -x -q 2>&1 | tail -3
echo "== sweep"
timeout 900 python scripts/spmv_sweep.py --oracle \
  "b128_s32:spmv_band_hot=128" "b128_s32_noov:spmv_band_hot=128,spmv_band_overlap=2" \
  "b96_s32:spmv_band_hot=96" "b64_s32:spmv_band_hot=64" "b48_s32:spmv_band_hot=48" "b192_s32:spmv_band_hot=192" \
  "b96_s48:spmv_band_hot=96,spmv_xcs_split=48" "b128_s48:spmv_band_hot=128,spmv_xcs_split=48" "b64_s64:spmv_band_hot=64,spmv_xcs_split=64" "b128_s24:spmv_band_hot=128,spmv_xcs_split=24" \
  "b128_s32_g16:spmv_band_hot=128,spmv_band_group=16" "b128_s32_g4:spmv_band_hot=128,spmv_band_group=4" \
  2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee $OUT/sweep.jsonl
for cfg in "b128_s32:spmv_band_hot=128,spmv_band_split_launch=1" "b128_s32_noov:spmv_band_hot=128,spmv_band_overlap=2,spmv_band_split_launch=1"; do
echo "== kernel trace, $cfg"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmv_sweep.py --steps 10 "$cfg" > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_seq.py $(find /tmp/st -name "*.db" | head -1) band_ ) 2>&1 | cut -c1-200
done | tee $OUT/kernel_seq.txt
} 2>&1 | tee $OUT/log.txt
