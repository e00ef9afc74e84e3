#!/bin/bash
mkdir -p gpurun_out/r01z
{
timeout 900 python -m pytest tests/test_spmv_gpu.py tests/test_spmm_gpu.py -m gpu -x -q 2>&1 | tail -3
for a in "" "--relabel 2" "--workload rmat1m" "--workload rmat1m --relabel 2" "--idx-bytes 4" "--idx-bytes 4 --relabel 2"; do
  echo "[$a]"; timeout 600 python bench.py --no-cpu-baseline --steps 30 $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['value'], d['roofline']['frac'])"
done
} 2>&1 | tee gpurun_out/r01z/log_relabel.txt
