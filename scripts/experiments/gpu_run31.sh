#!/bin/bash
mkdir -p gpurun_out/r01z
{
for a in "--permute-cols -65536" "--permute-cols -8192" "--permute-cols -524288"; do
  echo "[$a]"; timeout 600 python bench.py --no-cpu-baseline --steps 30 $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['value'])"
done
} 2>&1 | tee -a gpurun_out/r01z/log_permute_cols.txt
