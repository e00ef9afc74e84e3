#!/bin/bash
# round 4, final evidence B: whole-product parity of config 5 on the final sources, the default bench line and the spgemm5 line, kernel stats of the bench command, BiCGSTAB histories of the two flagged fuzz seeds
bash scripts/gpu_session.sh r11b spgemm_parity bench "bench:--workload spgemm5" stats
timeout 300 python scripts/bicgstab_history.py 57863 60527 2>&1 | grep -v amdgpu | cut -c1-20000 | tee gpurun_out/r11b/bicgstab_history.jsonl | cut -c1-300
