#!/bin/bash
mkdir -p gpurun_out/r01z
{
for a in "--workload rmat1m" "--workload rmat1m --xcs 1" "--workload rmat1m --xcs 1 --relabel 2" "--workload rmat:4000000:24" "--workload rmat:4000000:24 --xcs 2"; do
  echo "[$a]"; timeout 300 python bench.py --no-cpu-baseline --steps 40 $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['value'], d['roofline']['frac'])"
done
} 2>&1 | tee gpurun_out/r01z/log_rmat_small_sizes.txt
