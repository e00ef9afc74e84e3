#!/bin/bash
# config 2 (R-MAT 1M): what the hot kernel's 51 us are made of — developer build, stores / row starts switched off (wrong results on purpose)
mkdir -p gpurun_out/r11m
export TMPDIR=/tmp
export SPRS_HIP_LIBRARY=$GRAFT_REPO_ROOT/sprs_amd/libsprs_hip_dev.so
for cfg in "base" "nostore:spmv_band_debug=1" "norowstart:spmv_band_debug=4" "stream_only:spmv_band_debug=5"; do
  echo "== $cfg" | tee -a gpurun_out/r11m/config2_hot_ablation.txt
  ( cd /tmp && rm -rf /tmp/st && timeout 200 rocprofv3 --kernel-trace -d /tmp/st -o s -- python $GRAFT_REPO_ROOT/scripts/spmv_sweep.py --workload rmat1m --steps 10 "$cfg" > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/rocprof_seq.py $(find /tmp/st -name "*.db" | head -1) band_ spmv_ | sed -n '1,9p' ) 2>&1 | cut -c1-180 | tee -a gpurun_out/r11m/config2_hot_ablation.txt
done
