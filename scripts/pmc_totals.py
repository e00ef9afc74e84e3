#!/usr/bin/env python3
"""Turn a gpu_pmc.sh summary into per-SpMV totals: sums, over the kernels that make up ONE SpMV
(tile / sliced / carry / reduce kernels; plan-building kernels excluded), of the per-dispatch
counter means.  HBM traffic estimate per SpMV:
   read  = FETCH_SIZE [KiB] * 1024 * 2   (gfx950: requests are up to 128 B but tallied as 64 B —
                                         MI355X_MICROARCH.md, HBM section; upper estimate for gathers)
   write = WRITE_SIZE [KiB] * 1024
usage: pmc_totals.py <pmc_summary.txt> [out.json]
       pmc_totals.py <pmc_summary.txt> --update profiles/pmc_traffic.json <workload> <index_bytes> "<plan description>"
       (the second form replaces the entry of that workload / index width; the summary's "csrc_sha16:" line, written by
       scripts/gpu_pmc.sh on the GPU box, ties the numbers to the kernel sources they were measured on)"""
import json
import re
import sys

PER_SPMV = ("spmv_tile_kernel", "spmv_sliced_kernel", "spmv_carry_kernel", "spmv_sliced_carry_kernel",
            "xcs_reduce_kernel", "spmv_rowwave_kernel", "rl_permute_x_kernel",
            "band_permute_kernel", "band_gather_hot_kernel", "band_gather_kernel", "band_hot_kernel", "band_cold_kernel", "band_carry_kernel", "band_reduce_kernel",
            "band_tail_kernel")


def main():
    tot = {}
    per_kernel = {}
    # plan policy (round 5): a handle's first multiply runs on the plain tile index, so a profile of a banded-plan workload holds ONE
    # dispatch of spmv_tile_kernel / spmv_carry_kernel per handle beside the band_* kernels of the steady state: not part of "one SpMV"
    banded = any(line.startswith("sprs_hip::") and "band_hot_kernel" in line for line in open(sys.argv[1]))
    for line in open(sys.argv[1]):
        if not line.startswith("sprs_hip::"):
            continue
        parts = line.split()
        name = parts[0]
        full = line.replace("sprs_hip::", "").replace("(anonymous namespace)::", "")
        kern = re.sub(r"[<(].*", "", full.split()[0])
        if kern not in PER_SPMV:
            continue
        if banded and not kern.startswith("band_"):
            continue
        counter, n, mean = parts[-5], int(parts[-4]), float(parts[-3])
        inst = re.sub(r"\(.*", "", full).strip()          # with template arguments: band_cold_kernel<false, false> and <false, true> are two launches
        inst = re.sub(r"\s+\S+\s+\d+\s+\S+\s+\S+\s+\S+$", "", inst).strip() if "(" not in full else inst
        key = (inst, counter)
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
    if len(sys.argv) > 3 and sys.argv[2] == "--update":
        path, wl, ib, plan = sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
        sha = None
        for line in open(sys.argv[1]):
            if line.startswith("csrc_sha16:"):
                sha = line.split()[1]
        doc = json.load(open(path))
        doc["entries"] = [e for e in doc["entries"] if not (e["workload"] == wl and e["index_bytes"] == ib)]
        doc["entries"].insert(0, {"workload": wl, "index_bytes": ib, "plan": plan, "csrc_sha16": sha,
                                  "traffic_bytes": out.get("traffic_bytes_est"), "read_bytes": out.get("hbm_read_bytes_est"),
                                  "write_bytes": out.get("hbm_write_bytes_est"), "counters": tot,
                                  "per_kernel": out["per_kernel"]})
        open(path, "w").write(json.dumps(doc, indent=1) + "\n")
    elif len(sys.argv) > 2:
        open(sys.argv[2], "w").write(s + "\n")
    print(s)


if __name__ == "__main__":
    main()
