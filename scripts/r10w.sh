#!/bin/bash
# round 4, call w: Gauss-Seidel band schedule (chains of S rows per lane, hand-offs through LDS) — parity gate, then the 4096^2 heat system
bash scripts/gpu_session.sh r10w "gate:gauss_seidel"
timeout 300 python scripts/gauss_seidel_bench.py 4096 3 2>&1 | grep -v amdgpu | tee gpurun_out/r10w/gs_band.jsonl
GS_CHAIN=1 timeout 300 python scripts/gauss_seidel_bench.py 4096 3 2>&1 | grep -v amdgpu | tee -a gpurun_out/r10w/gs_band.jsonl
timeout 300 python scripts/gauss_seidel_bench.py 1024 3 2>&1 | grep -v amdgpu | tee -a gpurun_out/r10w/gs_band.jsonl
