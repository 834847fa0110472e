#!/bin/bash
# Rehearsal of bench.py's N = 2 code path on a ONE-GPU box: two ranks, both on cuda:0, gradients exchanged over gloo
# (RCCL needs one GPU per rank).  Checks that the multi-rank line is produced; the numbers mean nothing.
cd ${GRAFT_REPO_ROOT:-.}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 WORLD_SIZE=2 LOCAL_RANK=0 CREAM_DIST_BACKEND=gloo
RANK=1 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/rehearse_rank1.err &
RANK=0 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/rehearse_rank0.err | tail -1 | cut -c1-1500
wait
tail -3 gpurun_out/rehearse_rank0.err gpurun_out/rehearse_rank1.err
