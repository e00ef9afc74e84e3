#!/bin/bash
# round 4, call v: small banded plans — identity labels (no permutation launch), split 8, two rounds; config 2 warm and cold, config 3, kernel sequence
bash scripts/gpu_session.sh r10v "gate:test_spmv_band_gpu or test_spmv_gpu" "bench:--workload rmat1m --no-secondary" "bench:--workload rmat1m --cold-cache --no-secondary" "bench:--workload laplace4096 --no-secondary"
timeout 600 python scripts/spmv_sweep.py --workload rmat1m --steps 40 --repeat 2 "base" "noid:spmv_band_identity=2" "r4:spmv_band_rounds=4" "s24:spmv_band_split=24" 2>&1 | grep -v amdgpu | cut -c1-200 | tee gpurun_out/r10v/config2_sweep.jsonl
( cd /tmp && rm -rf /tmp/st && timeout 300 rocprofv3 --kernel-trace -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmv_sweep.py --workload rmat1m --steps 10 base > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_seq.py $(find /tmp/st -name "*.db" | head -1) band_ spmv_ ) 2>&1 | cut -c1-200 | tee gpurun_out/r10v/config2_kernel_sequence.txt | tail -12
