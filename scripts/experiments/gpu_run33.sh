#!/bin/bash
# PMC passes of the default bench command after the column relabelling went in (traffic per SpMV)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r01zr
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
: > $OUT/pmc_summary.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p$i.json 2> $OUT/p$i.err
  f=$(find /tmp/pmc_$i -name "*.db" | head -1)
  echo "== group $i: $grp" | tee -a $OUT/pmc_summary.txt
  if [ -n "$f" ]; then python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py "$f" sprs_hip | sed -n '/PMC counters/,$p' >> $OUT/pmc_summary.txt; else tail -3 $OUT/p$i.err; fi
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_totals.py $OUT/pmc_summary.txt $OUT/pmc_totals.json | tail -30
