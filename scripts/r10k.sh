#!/bin/bash
# round 4, call k: single-k rows streamed, hash kernel ranks by bins instead of sorting, new defaults (heavy 524288, wave window 2^15)
bash scripts/gpu_session.sh r10k "gate:test_spgemm_gpu" "spgemm_ab:base|SPGEMM_MIDWIN_SYM=14|SPGEMM_MID_KEEP=4|SPGEMM_MIDWIN=14" spgemm_stats
