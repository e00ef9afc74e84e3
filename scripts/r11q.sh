#!/bin/bash
mkdir -p gpurun_out/r11q
for m in plain k8first spmvplan aligned; do
  timeout 200 python scripts/spmm_timing_probe.py $m 2>&1 | grep -v amdgpu | tee -a gpurun_out/r11q/spmm_timing_probe.jsonl
done
timeout 200 python scripts/spmm_bench.py 10000000 32 8 16 2>&1 | grep -v amdgpu | tee -a gpurun_out/r11q/spmm_timing_probe.jsonl
timeout 200 python scripts/spmm_bench.py 10000000 32 16 2>&1 | grep -v amdgpu | tee -a gpurun_out/r11q/spmm_timing_probe.jsonl
