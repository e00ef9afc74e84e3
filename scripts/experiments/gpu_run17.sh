#!/bin/bash
TAG=${1:-r01q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/sgp_$i
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d /tmp/sgp_$i -o pmc -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 8 100 > $OUT/p$i.json 2> $OUT/p$i.err
  f=$(find /tmp/sgp_$i -name "*.db" | head -1)
  echo "== group $i: $grp"
  [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py "$f" large_ | sed -n '/PMC counters/,$p' | cut -c1-60,100-190
done 2>&1 | tee $OUT/pmc_spgemm.txt
