#!/bin/bash
# round 4, final evidence A: whole GPU suite + smoke; PMC traffic of the SpMV and of SpGEMM config 5 on the final sources; differential fuzzing
bash scripts/gpu_session.sh r11a tests pmc spgemm_traffic1 spgemm_stats
timeout 400 python scripts/fuzz_parity.py 300 50000 2>&1 | grep -v amdgpu | tail -4 | cut -c1-600 | tee gpurun_out/r11a/fuzz.jsonl
