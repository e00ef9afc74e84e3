#!/usr/bin/env python3
"""Register / LDS / occupancy table of the kernels of one .hip source (cross-compiles for gfx950, no GPU needed):
  python scripts/kernel_resources.py sprs_amd/csrc/spmv_band.hip [name-filter] [-DFLAG ...]
Parses hipcc's -Rpass-analysis=kernel-resource-usage remarks."""
import re
import subprocess
import sys

src = sys.argv[1]
flt = [a for a in sys.argv[2:] if not a.startswith("-")]
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "--cuda-device-only",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:") or body.startswith("Name:"):
        cur = {"name": body.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in body:
        k, v = body.split(":", 1)
        cur[k.strip()] = v.strip()
try:
    names = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
except OSError:
    names = [r["name"] for r in rows]
print("%-90s %5s %5s %5s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
for r, nm in zip(rows, names):
    nm = re.sub(r"\(.*", "", nm).replace("sprs_hip::(anonymous namespace)::", "").replace("void ", "")
    if flt and not any(f in nm for f in flt):
        continue
    print("%-90s %5s %5s %5s %7s %4s %7s" % (nm[:90], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", r.get("SGPRs", "?")),
                                           r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))
