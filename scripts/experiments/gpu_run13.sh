#!/bin/bash
TAG=${1:-r01m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== spgemm heavy-threshold sweep (config 5 and 300k)"
for h in 16384 65536 262144 1048576; do
  for cfg in "300000 8" "1000000 8"; do echo -n "heavy=$h $cfg: "; timeout 600 python tests/spgemm_bench.py $cfg 8 300 $h 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['seconds'], d['gflops'], d['parity']['structure_bit_exact'], d['parity']['values_bit_exact'])"; done
done
echo "== torchrun world_size 1 (nccl init path)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-300
} 2>&1 | tee $OUT/log.txt
