#!/bin/bash
mkdir -p gpurun_out/r11j
ROW_WEIGHT=8 timeout 600 python scripts/virtual_ranks.py 2>&1 | grep -v amdgpu | tee gpurun_out/r11j/virtual_ranks.json | cut -c1-3000
