#!/bin/bash
# round 4, call d: window width of the numeric wave kernel (one memory round trip per window: fewer, fatter windows)
bash scripts/gpu_session.sh r10d "gate:test_spgemm_gpu" "spgemm_ab:base|SPGEMM_MIDWIN=15|SPGEMM_MIDWIN=14|SPGEMM_MID=262144|SPGEMM_MID=1048576" spgemm_stats
