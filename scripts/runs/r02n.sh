#!/bin/bash
# round 2, GPU call n: reduce with 4 waves x 36 partials in flight, per-wave contiguous tables
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02n
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
echo "== pytest band"
timeout 900 python -m pytest tests/test_spmv_band_gpu.py -m gpu -x -q 2>&1 | tail -3
echo "== sweep"
timeout 900 python scripts/spmv_sweep.py --steps 30 --oracle "default:" "noov:spmv_band_overlap=2" "b96:spmv_band_hot=96" "b160:spmv_band_hot=160" "s24:spmv_xcs_split=24" "s48:spmv_xcs_split=48" "default_again:" "noov_again:spmv_band_overlap=2" \
  2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee $OUT/sweep.jsonl
for cfg in "noov:spmv_band_overlap=2"; do
echo "== kernel trace, $cfg"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmv_sweep.py --steps 10 "$cfg" > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_seq.py $(find /tmp/st -name "*.db" | head -1) band_ ) 2>&1 | cut -c1-200 | head -9
done | tee $OUT/kernel_seq.txt
} 2>&1 | tee $OUT/log.txt
