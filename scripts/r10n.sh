#!/bin/bash
# round 4, call n: the counting pass keeps the bitmaps of the large rows; the numeric workgroup kernel reads them back instead of walking for bits
bash scripts/gpu_session.sh r10n "gate:test_spgemm_gpu" "spgemm_ab:base|SPGEMM_KEEP_BITS=0|SPGEMM_WINLOG=16|SPGEMM_HEAVY=262144|SPGEMM_HEAVY=1048576" spgemm_stats spgemm_traffic1
