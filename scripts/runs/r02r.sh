#!/bin/bash
# round 2, GPU call r: SpGEMM config 5 — how the (window-major / row-major) task list is dealt to the XCDs
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02r
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
timeout 900 python - <<'PY'
import time, json, sys, torch
sys.path.insert(0, '.')
import sprs_amd
from sprs_amd import gen, smmp
from sprs_amd.device import DeviceCsMat
dev = torch.device("cuda", 0)
n = 1_000_000
indptr, indices, data = gen.rmat_csr(n, 8, device=dev, oversample=1.0)
a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
def t(f, reps=2):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0); del r
    return min(ts)
c = smmp.mul_csr_csr(a, a); del c
for order in (0, 2):
    for chunk in (0, 1, 4, 16, 64, 256, -1):
        sprs_amd.set_option("spgemm_task_order", order); sprs_amd.set_option("spgemm_xcd_chunk", chunk)
        print(json.dumps({"task_order": order, "xcd_chunk": chunk, "full_product_s": round(t(lambda: smmp.mul_csr_csr(a, a)), 4)}), flush=True)
PY
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
