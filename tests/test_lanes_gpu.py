"""sprs_amd/csrc/lanes.hpp on the hardware: the DPP / permlane-swap exchanges, the group scans and the group sorts that the
micro-row SpGEMM kernel is built from, each against plain index arithmetic / std::sort (scripts/probes/lane_ops.hip, a
stand-alone HIP program: it includes the shipped header).  The CPU emulator's model of the same controls is held to the same
probe in tests/test_emu_cpu.py, so "what the emulator runs" and "what the chip does" are tied to one reference."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE_SRC = os.path.join(ROOT, "scripts", "probes", "lane_ops.hip")
PROBE = os.path.join(ROOT, "scripts", "probes", "lane_ops.out")


@pytest.mark.gpu
def test_lane_exchanges_on_the_hardware():
    stale = not os.path.exists(PROBE) or os.path.getmtime(PROBE) < max(
        os.path.getmtime(PROBE_SRC), os.path.getmtime(os.path.join(ROOT, "sprs_amd", "csrc", "lanes.hpp")))
    if stale:
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        r = subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-o", PROBE, PROBE_SRC], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([PROBE, "1024"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] and all(v == 0 for v in out["mismatches"].values()), out
