#!/bin/bash
mkdir -p gpurun_out/r11s
timeout 200 python scripts/spmm_timing_probe.py like_bench 2>&1 | grep -v amdgpu | tee -a gpurun_out/r11s/spmm_timing_probe.jsonl
