#!/bin/bash
# round 2, GPU call j: reduce with XCD-contiguous row blocks; overlap x blocks-per-workgroup A/B (3 repeats of the interesting ones)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02j
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
echo "== sweep"
timeout 900 python scripts/spmv_sweep.py --steps 30 \
  "g16:" "g16_ov:spmv_band_overlap=1" "g8:spmv_band_group=8" "g8_ov:spmv_band_group=8,spmv_band_overlap=1" \
  "g12_ov:spmv_band_group=12,spmv_band_overlap=1" "g24_ov:spmv_band_group=24,spmv_band_overlap=1" "g32_ov:spmv_band_group=32,spmv_band_overlap=1" \
  "g16_ov_b96:spmv_band_overlap=1,spmv_band_hot=96" "g16_ov_b192:spmv_band_overlap=1,spmv_band_hot=192" "g16_ov_s48:spmv_band_overlap=1,spmv_xcs_split=48" "g16_ov_s24:spmv_band_overlap=1,spmv_xcs_split=24" \
  "g16_again:" "g16_ov_again:spmv_band_overlap=1" "g8_again:spmv_band_group=8" \
  2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee $OUT/sweep.jsonl
for cfg in "g16:" "g16_ov:spmv_band_overlap=1"; do
echo "== kernel trace, $cfg"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmv_sweep.py --steps 10 "$cfg" > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_seq.py $(find /tmp/st -name "*.db" | head -1) band_ ) 2>&1 | cut -c1-200
done | tee $OUT/kernel_seq.txt
} 2>&1 | tee $OUT/log.txt
