#!/bin/bash
# round 4, call o: the whole GPU suite + smoke on the current build; whole-product parity of config 5
bash scripts/gpu_session.sh r10o tests spgemm_parity
