#!/bin/bash
mkdir -p gpurun_out/r10z
export SPRS_HIP_LIBRARY=$PWD/sprs_amd/libsprs_hip_dev.so
for d in 0 2 4 8; do
  echo "-- gauss_seidel_debug=$d" | tee -a gpurun_out/r10z/log.txt
  GS_DEBUG=$d timeout 300 python scripts/gs_band_probe.py 4096 1 2>&1 | grep -v amdgpu | cut -c1-200 | tee -a gpurun_out/r10z/log.txt
done
