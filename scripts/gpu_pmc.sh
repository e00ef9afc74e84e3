#!/bin/bash
# PMC counter passes (one rocprofv3 run per counter group; --kernel-trace only, as gpurun requires).
# Usage: bash scripts/gpu_pmc.sh <tag> [bench args...]
TAG=${1:-pmc}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
[ -f $OUT/counters_list.txt ] || rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/p$i.json 2> $OUT/p$i.err
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  echo "== group $i: $grp -> $f"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    k = r.get("Kernel_Name", "")
    if "spmv_tile_kernel" in k or "spmv_rowwave" in k:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in acc.items():
    print("   %-32s n=%d mean=%.6g min=%.6g max=%.6g" % (c, len(v), sum(v)/len(v), min(v), max(v)))
PY
  # keep only the small CSVs
  find $OUT/p$i -name "*.db" -delete 2>/dev/null
done
