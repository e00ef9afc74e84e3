#!/bin/bash
# round 4, call u: config 2 (R-MAT 1M) — rounds / ranges / split / overlap of the small banded plan, once more on the round-4 build
mkdir -p gpurun_out/r10u
timeout 600 python scripts/spmv_sweep.py --workload rmat1m --steps 40 --repeat 3 "base" "r1:spmv_band_rounds=1" "r2:spmv_band_rounds=2" "r1run2:spmv_band_rounds=1,spmv_band_hot_run=2" "r2s8:spmv_band_rounds=2,spmv_band_split=8" "s8:spmv_band_split=8" "s8ov:spmv_band_split=8,spmv_band_overlap=1" "r2ov:spmv_band_rounds=2,spmv_band_overlap=1" "t8kr2:spmv_band_tile=8192,spmv_band_rounds=2" 2>&1 | grep -v amdgpu | cut -c1-260 | tee gpurun_out/r10u/config2_sweep.jsonl
