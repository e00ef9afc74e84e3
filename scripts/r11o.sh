#!/bin/bash
# software-pipelined reduction of the banded SpMV: gate, kernel sequence on R-MAT 10M and 1M, bench line
mkdir -p gpurun_out/r11o
timeout 900 python -m pytest tests/test_spmv_band_gpu.py tests/test_spmv_gpu.py -m gpu -x -q 2>&1 | tail -3
if [ "${PIPESTATUS[0]}" != 0 ]; then echo "gate failed"; exit 1; fi
bash scripts/gpu_session.sh r11o trace:base "bench:--no-secondary --no-cpu-baseline"
timeout 600 python scripts/spmv_sweep.py --workload rmat1m --steps 40 --repeat 2 --oracle "base" 2>&1 | grep -v amdgpu | cut -c1-260 | tee gpurun_out/r11o/config2.jsonl
