#!/bin/bash
# what the memory system delivers for the patterns that bound SpMV / SpMM (scripts/probes/hbm_patterns.hip); fabric traffic of the SpMM stream kernel (one counter per pass)
mkdir -p gpurun_out/r11g
timeout 300 scripts/probes/hbm_patterns.out 2>&1 | tee gpurun_out/r11g/hbm_patterns.jsonl
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt
  ( cd /tmp && timeout -s KILL 150 rocprofv3 --pmc $ctr --kernel-trace --kernel-include-regex "spmm_stream" -d /tmp/pt -o pmc -- python $GRAFT_REPO_ROOT/scripts/spmm_bench.py 10000000 32 8 16 > /dev/null 2>&1 )
  f=$(find /tmp/pt -name "*.db" | head -1)
  if [ -n "$f" ]; then python3 scripts/rocprof_summary.py "$f" sprs_hip | sed -n '/PMC counters/,$p' | cut -c1-250 | tee -a gpurun_out/r11g/spmm_traffic.txt; else echo "no db for $ctr" | tee -a gpurun_out/r11g/spmm_traffic.txt; fi
done
