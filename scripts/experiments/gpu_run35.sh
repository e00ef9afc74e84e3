#!/bin/bash
mkdir -p gpurun_out/r01z
{
for e in "" "BENCH_NO_RESORT=1"; do
  echo "[--permute-cols -1 --relabel 2 $e]"; env $e timeout 600 python bench.py --no-cpu-baseline --steps 30 --permute-cols -1 --relabel 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['value'], d['roofline']['frac'])"
done
} 2>&1 | tee gpurun_out/r01z/log_row_order.txt
