#!/bin/bash
mkdir -p gpurun_out/r11k
export TMPDIR=/tmp
for blk in "8 3" "4 2"; do
( cd /tmp && rm -rf /tmp/st && timeout 300 rocprofv3 --kernel-trace -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/block_spmv.py $blk 8 2>/dev/null | tail -1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_seq.py $(find /tmp/st -name "*.db" | head -1) band_ spmv_ ) 2>&1 | cut -c1-200 | tee -a gpurun_out/r11k/block_kernel_seq.txt
done
