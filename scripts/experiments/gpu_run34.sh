#!/bin/bash
mkdir -p gpurun_out/r01z
{
for a in "" ""; do
  echo "[half-octave classes $a]"; timeout 600 python bench.py --no-cpu-baseline --steps 30 $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['value'], d['roofline']['frac'])"
done
} 2>&1 | tee -a gpurun_out/r01z/log_relabel_sampled.txt
