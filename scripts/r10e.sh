#!/bin/bash
# round 4, call e: lane-order adds in the workgroup (large-row) kernel too
bash scripts/gpu_session.sh r10e "gate:test_spgemm_gpu" "spgemm_ab:base|SPGEMM_LANE_ORDER=2|SPGEMM_ORDERED=0|SPGEMM_TOKENS=2|SPGEMM_HEAVY=65536|SPGEMM_HEAVY=262144" spgemm_stats
