#!/bin/bash
mkdir -p gpurun_out/r10x
timeout 600 python scripts/gs_band_probe.py 4096 1 2 8 64 2>&1 | grep -v amdgpu | tee gpurun_out/r10x/gs_band_probe.jsonl
