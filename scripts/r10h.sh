#!/bin/bash
# round 4, call h: every load of the B entries and of the window edges unconditional (no per-load waits); chunk sizes
bash scripts/gpu_session.sh r10h "gate:test_spgemm_gpu" "spgemm_ab:SPGEMM_HEAVY=524288|SPGEMM_HEAVY=524288 SPGEMM_MID_KEEP=4|SPGEMM_HEAVY=524288 SPGEMM_MID_KEEP_SYM=16|SPGEMM_HEAVY=524288 SPGEMM_MIDWIN_SYM=14|SPGEMM_HEAVY=524288 SPGEMM_MIDWIN=15|SPGEMM_HEAVY=524288 SPGEMM_MID=262144" spgemm_stats
