#!/bin/bash
# Rehearsal of bench.py's process-per-GPU mode on ONE device: two ranks that share device 0, collectives staged through the host
# (--backend gloo; nccl = RCCL needs one device per rank).  What it shows is the bench LINE of that mode -- `sharded_step` with the
# per-rank stages gathered through torch.distributed, streamed_ranks, exchange bytes -- not a speed.
#   tools/rehearse_torchrun.sh [n] [m]      -> gpurun_out/rehearse_torchrun.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-768}; M=${2:-512}
mkdir -p $R/gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 WORLD_SIZE=2 LOCAL_RANK=0 HDM_BENCH_NO_PROBE=1
RANK=1 python3 $R/bench.py --gpus 2 --backend gloo --n $N --m $M --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/rehearse_torchrun_rank1.log 2>&1 &
P1=$!
RANK=0 python3 $R/bench.py --gpus 2 --backend gloo --n $N --m $M --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/rehearse_torchrun.json 2> $R/gpurun_out/rehearse_torchrun_rank0.err
RC=$?
wait $P1
tail -c 2500 $R/gpurun_out/rehearse_torchrun.json
exit $RC
