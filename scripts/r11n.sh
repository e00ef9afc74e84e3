#!/bin/bash
# hot kernel with two tiles in flight per wave (spmv_band_prefetch): gate, then config 2, a rank block of an 8-way cut, and the 10M matrix, 1 against 2
mkdir -p gpurun_out/r11n
timeout 900 python -m pytest tests/test_spmv_band_gpu.py tests/test_spmv_gpu.py -m gpu -x -q 2>&1 | tail -3
if [ "${PIPESTATUS[0]}" != 0 ]; then echo "gate failed"; exit 1; fi
timeout 600 python scripts/spmv_sweep.py --workload rmat1m --steps 40 --repeat 3 --oracle "pf2" "pf1:spmv_band_prefetch=1" 2>&1 | grep -v amdgpu | cut -c1-260 | tee gpurun_out/r11n/config2_prefetch.jsonl
timeout 600 python scripts/spmv_sweep.py --row-block 3/8 --steps 30 --repeat 3 "pf2" "pf1:spmv_band_prefetch=1" 2>&1 | grep -v amdgpu | cut -c1-260 | tee gpurun_out/r11n/block8_prefetch.jsonl
timeout 600 python scripts/spmv_sweep.py --steps 30 --repeat 3 --oracle "pf1" "pf2:spmv_band_prefetch=2" 2>&1 | grep -v amdgpu | cut -c1-260 | tee gpurun_out/r11n/rmat10m_prefetch.jsonl
