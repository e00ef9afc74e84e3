#!/usr/bin/env python3
"""Turn a gpu_pmc.sh summary into per-SpMV totals: sums, over the kernels that make up ONE SpMV
(tile / sliced / carry / reduce kernels; plan-building kernels excluded), of the per-dispatch
counter means.  HBM traffic estimate per SpMV:
   read  = FETCH_SIZE [KiB] * 1024 * 2   (gfx950: requests are up to 128 B but tallied as 64 B —
                                         MI355X_MICROARCH.md, HBM section; upper estimate for gathers)
   write = WRITE_SIZE [KiB] * 1024
usage: pmc_totals.py <pmc_summary.txt> [out.json]"""
import json
import re
import sys

PER_SPMV = ("spmv_tile_kernel", "spmv_sliced_kernel", "spmv_carry_kernel", "spmv_sliced_carry_kernel",
            "xcs_reduce_kernel", "spmv_rowwave_kernel", "rl_permute_x_kernel")


def main():
    tot = {}
    per_kernel = {}
    for line in open(sys.argv[1]):
        if not line.startswith("sprs_hip::"):
            continue
        parts = line.split()
        name = parts[0]
        kern = re.sub(r"[<(].*", "", name.replace("sprs_hip::", ""))
        if kern not in PER_SPMV:
            continue
        counter, n, mean = parts[-5], int(parts[-4]), float(parts[-3])
        key = (kern, counter)
        if key in per_kernel:
            continue                      # the summary repeats groups; first wins
        per_kernel[key] = mean
        tot[counter] = tot.get(counter, 0.0) + mean
    out = {"counters_per_spmv": tot,
           "per_kernel": {"%s:%s" % k: v for k, v in sorted(per_kernel.items())}}
    if "FETCH_SIZE" in tot:
        out["hbm_read_bytes_est"] = tot["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in tot:
        out["hbm_write_bytes_est"] = tot["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        out["traffic_bytes_est"] = out["hbm_read_bytes_est"] + out["hbm_write_bytes_est"]
    s = json.dumps(out, indent=1, sort_keys=True)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(s + "\n")
    print(s)


if __name__ == "__main__":
    main()
