#!/bin/bash
# fabric traffic of the final SpMM kernels (re-laid-out rhs copy, stream, fix-up), one counter per pass; and a sweep over k with the copy on / off
mkdir -p gpurun_out/r12b
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt
  ( cd /tmp && timeout -s KILL 150 rocprofv3 --pmc $ctr --kernel-trace --kernel-include-regex "spmm_" -d /tmp/pt -o pmc -- python $GRAFT_REPO_ROOT/scripts/spmm_bench.py 10000000 32 8 16 32 > /dev/null 2>&1 )
  f=$(find /tmp/pt -name "*.db" | head -1)
  if [ -n "$f" ]; then python3 scripts/rocprof_summary.py "$f" sprs_hip | sed -n '/PMC counters/,$p' | cut -c1-250 | tee -a gpurun_out/r12b/spmm_traffic.txt; else echo "no db for $ctr" | tee -a gpurun_out/r12b/spmm_traffic.txt; fi
done
for o in "spmm_relayout=2" ""; do
  echo "-- ${o:-auto (copy on)}" | tee -a gpurun_out/r12b/spmm_k_sweep.jsonl
  SPRS_OPTS="$o" timeout 300 python scripts/spmm_bench.py 10000000 32 2 4 8 12 16 24 32 48 64 100 2>&1 | grep -v amdgpu | cut -c1-130 | tee -a gpurun_out/r12b/spmm_k_sweep.jsonl
done
