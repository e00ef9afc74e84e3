#!/usr/bin/env python3
"""HBM traffic of ONE SpGEMM (config 5) from a rocprofv3 --pmc pass over tests/spgemm_bench.py (which runs two products):
sum over every sprs_hip kernel of dispatches x mean(FETCH_SIZE / WRITE_SIZE), divided by the number of products;
read = FETCH_SIZE [KiB] * 1024 * 2 (gfx950 tallies 128-B requests as 64 B: MI355X_MICROARCH.md, HBM section), write =
WRITE_SIZE [KiB] * 1024.
usage: spgemm_traffic.py <summary.txt> <products_in_run> [--update profiles/pmc_traffic.json <index_bytes> <csrc_sha16>]"""
import json
import sys


def main():
    tot = {}
    per_kernel = {}
    for line in open(sys.argv[1]):
        if not line.startswith("sprs_hip::"):
            continue
        parts = line.split()
        counter, n, mean = parts[-5], int(parts[-4]), float(parts[-3])
        if counter not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        kern = line.replace("sprs_hip::", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
        tot[counter] = tot.get(counter, 0.0) + n * mean
        per_kernel["%s:%s" % (kern, counter)] = per_kernel.get("%s:%s" % (kern, counter), 0.0) + n * mean
    products = int(sys.argv[2])
    read = tot.get("FETCH_SIZE", 0.0) * 1024 * 2 / products
    write = tot.get("WRITE_SIZE", 0.0) * 1024 / products
    out = {"read_bytes": read, "write_bytes": write, "traffic_bytes": read + write, "products_in_run": products,
           "per_kernel_KiB_whole_run": per_kernel}
    print(json.dumps(out, indent=1, sort_keys=True))
    if len(sys.argv) > 3 and sys.argv[3] == "--update":
        path, ib, sha = sys.argv[4], int(sys.argv[5]), sys.argv[6]
        doc = json.load(open(path))
        doc["entries"] = [e for e in doc["entries"] if not (e["workload"] == "spgemm5" and e["index_bytes"] == ib)]
        doc["entries"].append({"workload": "spgemm5", "index_bytes": ib, "csrc_sha16": sha,
                               "plan": "one sprs_hip_spgemm_f64 call on BASELINE config 5 (all kernels: plan, symbolic, scans, numeric)",
                               "traffic_bytes": read + write, "read_bytes": read, "write_bytes": write,
                               "per_kernel_KiB_whole_run": per_kernel})
        open(path, "w").write(json.dumps(doc, indent=1) + "\n")


if __name__ == "__main__":
    main()
