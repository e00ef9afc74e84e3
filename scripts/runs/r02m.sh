#!/bin/bash
# round 2, GPU call m: is the final reduction bound by memory?  (partials forced into a 512 KiB window: wrong results, timing only)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02m
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
for cfg in "default:spmv_band_overlap=2" "reduce_in_cache:spmv_band_overlap=2,spmv_xmask=65535"; do
echo "== kernel trace, $cfg"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmv_sweep.py --steps 10 "$cfg" > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_seq.py $(find /tmp/st -name "*.db" | head -1) band_ ) 2>&1 | cut -c1-200 | head -9
done | tee $OUT/kernel_seq.txt
} 2>&1 | tee $OUT/log.txt
