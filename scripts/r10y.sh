#!/bin/bash
# what scales the step time of the band kernel: waves per workgroup
mkdir -p gpurun_out/r10y
for W in 4 16 2; do
  sed -i "s/constexpr int GB_WAVES = [0-9]*, /constexpr int GB_WAVES = $W, /" sprs_amd/csrc/gauss_seidel.hip
  make -s -C sprs_amd/csrc 2>&1 | grep -E "error"
  echo "-- GB_WAVES=$W" | tee -a gpurun_out/r10y/log.txt
  timeout 300 python scripts/gs_band_probe.py 4096 1 2>&1 | grep -v amdgpu | tee -a gpurun_out/r10y/log.txt
done
