#!/bin/bash
# round 2, GPU call q: SpGEMM plan / task order — whole GPU suite, spgemm5 bench line, task-order A/B, numeric-on-plan time, kernel trace
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02q
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -14
echo "== bench spgemm5"
timeout 900 python bench.py --workload spgemm5 --steps 4 2>/dev/null | tee $OUT/bench_spgemm5.json
echo "== plan timings / task order"
timeout 600 python - <<'PY'
import time, json, sys, torch
sys.path.insert(0, '.')
import sprs_amd
from sprs_amd import gen, smmp
from sprs_amd.device import DeviceCsMat
dev = torch.device("cuda", 0)
n = 1_000_000
indptr, indices, data = gen.rmat_csr(n, 8, device=dev, oversample=1.0)
a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
def t(f, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return r, min(ts), ts
for order in (0, 2):
    sprs_amd.set_option("spgemm_task_order", order)
    c, best, ts = t(lambda: smmp.mul_csr_csr(a, a))
    print(json.dumps({"task_order": order, "full_product_s": round(best, 4), "all": [round(x, 4) for x in ts]}))
    del c
sprs_amd.set_option("spgemm_task_order", 0)
plan, tp, _ = t(lambda: smmp.SpgemmPlan(a, a), reps=2)
c, tprod, _ = t(lambda: plan.product(), reps=2)
_, tnum, ts = t(lambda: plan.numeric(c))
print(json.dumps({"plan_create_s": round(tp, 4), "plan_product_s": round(tprod, 4), "plan_numeric_s": round(tnum, 4), "numeric_runs": [round(x, 4) for x in ts]}))
PY
echo "== kernel trace spgemm"
( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/tests/spgemm_bench.py 1000000 8 8 1 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/st -name "*.db" | head -1) sprs_hip ) 2>&1 | grep -E "^kernel|sprs_hip" | cut -c1-200 | head -24
} 2>&1 | tee $OUT/log.txt
