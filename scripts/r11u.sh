#!/bin/bash
# round 4, last: the other configs' bench lines on the final build, and 300 s of differential fuzzing (all kinds, SpMM in its three modes)
mkdir -p gpurun_out/r11u
for w in "--workload rmat1m" "--workload rmat1m --cold-cache" "--workload laplace4096" "--workload spgemm5"; do
  timeout 600 python bench.py $w 2>/dev/null | tee -a gpurun_out/r11u/bench_lines.jsonl | cut -c1-400
done
timeout 400 python scripts/fuzz_parity.py 300 70000 2>&1 | grep -v amdgpu | tail -4 | cut -c1-700 | tee gpurun_out/r11u/fuzz.jsonl
