#!/bin/bash
# SpGEMM v3 PMC passes on config 5 (winlog 17): what saturates the CU?
OUT=$GRAFT_REPO_ROOT/gpurun_out/r01z
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
: > $OUT/pmc_spgemm_v3.txt
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  SPGEMM_WINLOG=17 timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 8 1 > $OUT/p$i.json 2> $OUT/p$i.err
  f=$(find /tmp/pmc_$i -name "*.db" | head -1)
  echo "== group $i: $grp" | tee -a $OUT/pmc_spgemm_v3.txt
  if [ -n "$f" ]; then python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py "$f" large_ | sed -n '/PMC counters/,$p' | cut -c41-200 | tee -a $OUT/pmc_spgemm_v3.txt; else tail -3 $OUT/p$i.err; fi
done
