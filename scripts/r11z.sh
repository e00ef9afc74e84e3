#!/bin/bash
# round 4, final session on the sources as committed: GPU suite + smoke, the driver's bench command, kernel stats of it, traffic PMC of the SpMV and of one config-5 product, whole-product parity, SpMM bench
bash scripts/gpu_session.sh r11z tests bench stats pmc spgemm_traffic1 spgemm_stats spgemm_parity
timeout 300 python scripts/spmm_bench.py 10000000 32 8 16 32 64 2>&1 | grep -v amdgpu | tee gpurun_out/r11z/spmm_bench.jsonl
