#!/bin/bash
# differential fuzzing on the final build with the re-laid-out rhs copy among the SpMM / dense-dispatch modes
mkdir -p gpurun_out/r11zi
timeout 300 python scripts/fuzz_parity.py 200 90000 2>&1 | grep -v amdgpu | tail -4 | cut -c1-700 | tee gpurun_out/r11zi/fuzz.jsonl
