#!/bin/bash
# round 2, GPU call b: hot kernel as per-wave pipelines — parity, sweep, traces with the short rows launched apart
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02b
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
echo "== pytest band"
timeout 900 python -m pytest tests/test_spmv_band_gpu.py -m gpu -x -q 2>&1 | tail -4
echo "== sweep"
SPRS_HIP_DEBUG=1 timeout 900 python scripts/spmv_sweep.py --oracle \
  "band24:" "old_sliced:spmv_band=2" \
  "band16:spmv_band_hot=16" "band32:spmv_band_hot=32" "band48:spmv_band_hot=48" "band64:spmv_band_hot=64" "band96:spmv_band_hot=96" \
  "b24_s64:spmv_xcs_split=64" "b48_s64:spmv_band_hot=48,spmv_xcs_split=64" "b64_s64:spmv_band_hot=64,spmv_xcs_split=64" "b96_s64:spmv_band_hot=96,spmv_xcs_split=64" \
  "b64_s128:spmv_band_hot=64,spmv_xcs_split=128" "b96_s128:spmv_band_hot=96,spmv_xcs_split=128" \
  "b48_ph4:spmv_band_hot=48,spmv_band_phases=4" "b48_ph8:spmv_band_hot=48,spmv_band_phases=8" \
  "b48_g1:spmv_band_hot=48,spmv_band_group=1" "b48_g2:spmv_band_hot=48,spmv_band_group=2" "b48_g8:spmv_band_hot=48,spmv_band_group=8" \
  2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.jsonl
for cfg in "band24:spmv_band_split_launch=1" "b64_s64:spmv_band_hot=64,spmv_xcs_split=64,spmv_band_split_launch=1"; do
echo "== kernel trace, $cfg"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmv_sweep.py --steps 10 "$cfg" > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|^#|band_|permute" | cut -c1-200
done | tee $OUT/kernel_stats.txt
} 2>&1 | tee $OUT/log.txt
