#!/bin/bash
# round 4, call q: wave-per-row kernel stores its indices staged through LDS, one coalesced store per pass
bash scripts/gpu_session.sh r10q "gate:test_spgemm_gpu" "spgemm_ab:base|SPGEMM_MID_KEEP=4|SPGEMM_MIDWIN=14" spgemm_stats spgemm_traffic1
