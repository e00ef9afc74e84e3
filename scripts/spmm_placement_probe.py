#!/usr/bin/env python3
"""Does the time of the SpMM call depend on WHERE its dense operands sit?  rhs and out are views into one 8 GiB buffer at
chosen offsets; the matrix, the plan and the library's carry buffer stay where they are."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprs_amd import _ffi, gen                      # noqa: E402
from sprs_amd.device import DeviceCsMat             # noqa: E402


def main():
    n, k = 10_000_000, 16
    dev = torch.device("cuda", 0)
    indptr, indices, data = gen.rmat_csr(n, 32, device=dev)
    a = DeviceCsMat.wrap_torch((n, n), indptr, indices, data)
    pool = torch.empty(1 << 30, dtype=torch.float64, device=dev)          # 8 GiB
    src = gen.dense_vector(n * k, seed=5, device=dev)
    nb = n * k                                                             # doubles per operand

    def measure(rhs_off, out_off):
        rhs = pool[rhs_off:rhs_off + nb]
        out = pool[out_off:out_off + nb]
        rhs.copy_(src)
        call = lambda: _ffi.check(_ffi.lib.sprs_hip_spmm_rowmaj_f64(a._h, C.c_void_p(rhs.data_ptr()), n, k, k, C.c_void_p(out.data_ptr()), n, k, 0, None))
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for p, q in evs:
            p.record()
            call()
            q.record()
        torch.cuda.synchronize()
        ms = sum(p.elapsed_time(q) for p, q in evs) / len(evs)
        print(json.dumps({"rhs_off_bytes": rhs_off * 8, "out_off_bytes": out_off * 8, "gap_bytes": (out_off - rhs_off - nb) * 8, "ms": round(ms, 3),
                          "rhs_ptr": hex(rhs.data_ptr()), "out_ptr": hex(out.data_ptr())}), flush=True)

    print(json.dumps({"indices": hex(indices.data_ptr()), "data": hex(data.data_ptr()), "pool": hex(pool.data_ptr())}))
    for gap in (0, 512, 4096 // 8 * 8, 65536, 1 << 20, (1 << 21), (1 << 21) + 4096, 1 << 24, 1 << 28, 1 << 30, 3 << 30):
        measure(0, nb + gap // 8)
    for off in (512, 4096, 1 << 20, 1 << 26):
        measure(off // 8, (1 << 29) + off // 8)                            # both shifted, 4 GiB apart
    measure(1 << 29, 0)                                                    # out below rhs


if __name__ == "__main__":
    main()
