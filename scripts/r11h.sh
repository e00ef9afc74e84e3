#!/bin/bash
mkdir -p gpurun_out/r11h
timeout 300 scripts/probes/hbm_read_variants.out 2>&1 | tee gpurun_out/r11h/hbm_read_variants.jsonl
