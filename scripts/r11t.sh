#!/bin/bash
mkdir -p gpurun_out/r11t
cd scripts && timeout 300 python spmm_tlb_probe.py 2>&1 | grep -v amdgpu | tee ../gpurun_out/r11t/spmm_tlb_probe.jsonl
