#!/bin/bash
mkdir -p gpurun_out/r11r
timeout 300 python scripts/spmm_placement_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r11r/spmm_placement_probe.jsonl
