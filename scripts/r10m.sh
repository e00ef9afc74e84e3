#!/bin/bash
# round 4, call m: workgroup kernel without the retain path (220 -> 80 B of scratch), indices staged through LDS and stored coalesced
bash scripts/gpu_session.sh r10m "gate:test_spgemm_gpu" "spgemm_ab:base|SPGEMM_OCCUPANCY=2|SPGEMM_TOKENS=2|SPGEMM_ORDERED=0" spgemm_stats spgemm_traffic1
