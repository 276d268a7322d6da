#!/bin/bash
# round 3: SyncBatchNorm mailbox -- tests, 1-GPU A/B (no sync / loop-back mailbox), 2 ranks on one GPU (mailbox vs all-reduce)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_syncbn.py tests/test_gpu_ddp.py -x -q 2>&1 | tail -15 | tee gpurun_out/r03_syncbn_tests.log
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | tee -a gpurun_out/r03_syncbn_ab.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --sync-bn 2>&1 | tail -1 | tee -a gpurun_out/r03_syncbn_ab.log
done
for t in mailbox rccl; do
TCVOM_SYNCBN=$t TCVOM_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 6 --warmup 2 --no-profile 2>&1 | tail -2 | tee -a gpurun_out/r03_syncbn_2rank.log
done
