#!/bin/bash
# SpGEMM v3 (entry-parallel expansion, LDS accumulators with order tags, window size templated): parity + sweep
mkdir -p gpurun_out/r01z
{
timeout 900 python -m pytest tests/test_spgemm_gpu.py -m gpu -x -q -k "not config5" 2>&1 | tail -5
for wl in 19 18 17 16; do for hv in 16384 65536; do
  echo "--- winlog $wl heavy $hv"
  SPGEMM_WINLOG=$wl SPGEMM_HEAVY=$hv timeout 300 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep -v amdgpu.ids
done; done
echo "--- prof winlog 19"
SPGEMM_PROF=1 SPGEMM_WINLOG=19 timeout 300 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep spgemm_prof
echo "--- prof winlog 17"
SPGEMM_PROF=1 SPGEMM_WINLOG=17 timeout 300 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep spgemm_prof
} 2>&1 | tee gpurun_out/r01z/log_v3b.txt
