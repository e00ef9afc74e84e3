#!/bin/bash
# SpGEMM config 5 on P A P^T: does the relabelled product move less data (an algorithmic effect) or the same (an address effect)?  FETCH_SIZE, one product
mkdir -p gpurun_out/r12e
export TMPDIR=/tmp
for p in 7; do
  rm -rf /tmp/pt
  ( cd /tmp && SPGEMM_PERMUTE=$p timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "sprs_hip" -d /tmp/pt -o pmc -- python $GRAFT_REPO_ROOT/scripts/spgemm_one.py 1 > /dev/null 2>&1 )
  f=$(find /tmp/pt -name "*.db" | head -1)
  echo "== SPGEMM_PERMUTE=$p" | tee -a gpurun_out/r12e/spgemm_permuted_fetch.txt
  if [ -n "$f" ]; then python3 scripts/rocprof_summary.py "$f" sprs_hip | sed -n '/PMC counters/,$p' | grep -E "rows_kernel|kernel " | cut -c1-200 | tee -a gpurun_out/r12e/spgemm_permuted_fetch.txt; fi
done
