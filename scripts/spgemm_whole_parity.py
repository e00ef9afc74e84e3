#!/usr/bin/env python3
"""WHOLE-product parity of BASELINE config 5 (C = A*A, R-MAT 1M x 1M ~8 nnz/row; 3.3e9 stored entries, 53 GB): every row
block of the GPU result against the CPU oracle (the C restatement of smmp::mul_csr_csr, sprs/src/sparse/smmp.rs:196-416,
OpenMP over the block's rows) — indptr and indices bit for bit, values bit for bit (and their relative error, should the
bits ever differ).  The GPU product is computed once and stays in HBM; the oracle streams over it block by block.
Writes one JSON record (kept under profiles/).
usage: spgemm_whole_parity.py <out.json> [block_rows=20000] [idx_bytes=8]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sprs_amd import gen, smmp                      # noqa: E402
from sprs_amd.device import DeviceCsMat              # noqa: E402


def main():
    out_path = sys.argv[1]
    block = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    idx_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    from oracle import oracle                        # the checker
    dev = torch.device("cuda", 0)
    n = 1_000_000
    idt = torch.int64 if idx_bytes == 8 else torch.int32
    indptr, indices, data = gen.rmat_csr(n, 8, device=dev, idx_dtype=idt, ptr_dtype=torch.int64, oversample=1.0)
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    t0 = time.perf_counter()
    c = smmp.mul_csr_csr(a, a)
    torch.cuda.synchronize()
    gpu_s = time.perf_counter() - t0
    npi = np.uint64 if idx_bytes == 8 else np.uint32
    ip_h = indptr.cpu().numpy().view(np.uint64)
    ix_h = indices.cpu().numpy().view(npi)
    dt_h = data.cpu().numpy()
    rows_ok = idx_ok = val_bits_ok = True
    worst, entries, cpu_s, bad_blocks = 0.0, 0, 0.0, []
    sum_ix = sum_dt = 0                                  # 64-bit wrap-around sums of the oracle's indices / value bits (cheap fingerprints)
    nblocks = 0
    for r0 in range(0, n, block):
        r1 = min(n, r0 + block)
        s0, e0 = int(ip_h[r0]), int(ip_h[r1])
        t = time.perf_counter()
        _, rip, rix, rdt = oracle.mul_csr_csr((r1 - r0, n), ip_h[r0:r1 + 1], ix_h[s0:e0], dt_h[s0:e0], (n, n), ip_h, ix_h, dt_h, threads=0)
        cpu_s += time.perf_counter() - t
        gip, gix, gdt = c.slice_outer_to_host(r0, r1)
        gip = gip - gip[0]
        ok_p = bool(np.array_equal(gip, rip))
        ok_i = ok_p and bool(np.array_equal(gix, rix))
        ok_v = ok_i and bool(np.array_equal(gdt.view(np.uint64), rdt.view(np.uint64)))
        if ok_i and not ok_v and rdt.size:
            worst = max(worst, float(np.max(np.abs(gdt - rdt) / np.maximum(np.abs(rdt), 1e-300))))
        rows_ok &= ok_p
        idx_ok &= ok_i
        val_bits_ok &= ok_v
        if not ok_v:
            bad_blocks.append(r0)
        entries += int(rix.size)
        sum_ix = (sum_ix + int(np.sum(rix.astype(np.uint64), dtype=np.uint64))) & 0xFFFFFFFFFFFFFFFF
        sum_dt = (sum_dt + int(np.sum(rdt.view(np.uint64), dtype=np.uint64))) & 0xFFFFFFFFFFFFFFFF
        nblocks += 1
    import bench
    rec = {"workload": "BASELINE config 5: C = A*A, R-MAT 1M x 1M ~8 nnz/row (seed 1, no oversampling)", "index_bytes": idx_bytes,
           "rows": n, "nnz_a": int(ix_h.size), "nnz_c_gpu": int(c.nnz()), "entries_compared": entries, "row_blocks": nblocks,
           "block_rows": block, "indptr_bit_exact": rows_ok, "indices_bit_exact": idx_ok, "values_bit_exact": val_bits_ok,
           "max_rel_err_where_bits_differ": worst, "tolerance": 1e-10, "blocks_with_differences": bad_blocks[:20],
           "ok": bool(rows_ok and idx_ok and worst <= 1e-10 and entries == int(c.nnz())),
           "oracle": "oracle/sprs_oracle_impl.h mul_csr_csr (smmp.rs:196-416), ThreadingStrategy::Automatic, %d host threads" % oracle.num_procs(),
           "oracle_seconds_all_blocks": round(cpu_s, 2), "gpu_seconds_first_product": round(gpu_s, 4),
           "sum64_oracle_indices": sum_ix, "sum64_oracle_value_bits": sum_dt, "csrc_sha16": bench.csrc_sha16("spgemm")}
    with open(out_path, "w") as f:
        f.write(json.dumps(rec, indent=1) + "\n")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
