#!/bin/bash
# SpGEMM config 5, natural labels (control) against P A P^T: which kernels are faster on the relabelled matrix?
mkdir -p gpurun_out/r11zf
export TMPDIR=/tmp
for p in 0 7; do
  echo "== SPGEMM_PERMUTE=$p" | tee -a gpurun_out/r11zf/spgemm_permuted_kernel_stats.txt
  ( cd /tmp && rm -rf /tmp/st && SPGEMM_PERMUTE=$p timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 8 1 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|rows_kernel" | cut -c1-200 | tee -a gpurun_out/r11zf/spgemm_permuted_kernel_stats.txt
done
