#!/bin/bash
# round 2, GPU call k: short rows tiled (8192 hottest x entries in LDS), reduce with 8 waves per row block
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02k
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
echo "== pytest band"
timeout 900 python -m pytest tests/test_spmv_band_gpu.py -m gpu -x -q 2>&1 | tail -3
echo "== sweep"
timeout 900 python scripts/spmv_sweep.py --steps 30 --oracle \
  "default:" "short_gather:spmv_band_short=2" "default_ov:spmv_band_overlap=1" "g8:spmv_band_group=8" "g8_ov:spmv_band_group=8,spmv_band_overlap=1" \
  "s24:spmv_xcs_split=24" "s48:spmv_xcs_split=48" "s64:spmv_xcs_split=64" "s96:spmv_xcs_split=96" "b96:spmv_band_hot=96" "b96_s64:spmv_band_hot=96,spmv_xcs_split=64" "b64_s64:spmv_band_hot=64,spmv_xcs_split=64" \
  "default_again:" "short_gather_again:spmv_band_short=2" \
  2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee $OUT/sweep.jsonl
for cfg in "default:" "s64:spmv_xcs_split=64"; do
echo "== kernel trace, $cfg"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmv_sweep.py --steps 10 "$cfg" > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_seq.py $(find /tmp/st -name "*.db" | head -1) band_ ) 2>&1 | cut -c1-200
done | tee $OUT/kernel_seq.txt
} 2>&1 | tee $OUT/log.txt
