#!/bin/bash
mkdir -p gpurun_out/r11l
for blk in "8 0" "8 3" "8 7" "4 2"; do
  for o in "" "spmv_band_overlap=1"; do
    SPRS_OPTS="$o" timeout 200 python scripts/block_spmv.py $blk 30 2>&1 | grep -v amdgpu | tail -1 | tee -a gpurun_out/r11l/block_overlap_ab.jsonl
  done
done
