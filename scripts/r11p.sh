#!/bin/bash
mkdir -p gpurun_out/r11p
timeout 400 python scripts/spmm_timing_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r11p/spmm_timing_probe.jsonl
