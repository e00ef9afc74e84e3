#!/bin/bash
# PMC counter passes (one rocprofv3 run per counter group; --kernel-trace only, as gpurun requires).
# Usage: bash scripts/gpu_pmc.sh <tag> [bench args...]   -> gpurun_out/<tag>/pmc_summary.txt
TAG=${1:-pmc}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; print('csrc_sha16:', bench.csrc_sha16('spmv'))" > $OUT/pmc_summary.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"; do
  i=$((i+1))
  if [ -n "$PMC_ONLY" ] && [ $i -ne $PMC_ONLY ]; then continue; fi
  if [ $i -gt ${PMC_GROUPS:-6} ]; then break; fi
  rm -rf /tmp/pmc_$i
  timeout -s KILL 240 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/p$i.json 2> $OUT/p$i.err
  f=$(find /tmp/pmc_$i -name "*.db" | head -1)
  echo "== group $i: $grp" | tee -a $OUT/pmc_summary.txt
  if [ -n "$f" ]; then python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py "$f" sprs_hip | sed -n '/PMC counters/,$p' | tee -a $OUT/pmc_summary.txt; else tail -3 $OUT/p$i.err; fi
done
