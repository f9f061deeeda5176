#!/bin/bash
# two ranks on the ONE GPU of the test box: RCCL refuses the duplicate device -> exercises bench.py's multi-rank flow with
# the file-barrier fallback (rendezvous, barriers, max-over-ranks timing, rank 0 prints the JSON line)
cd $GRAFT_REPO_ROOT
export WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29540 LOCAL_RANK=0 XH_RENDEZVOUS_KEY=two_ranks_$$
RANK=1 timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --grid 365x720x720 > gpurun_out/two_r1.out 2> gpurun_out/two_r1.err &
RANK=0 timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --grid 365x720x720 > gpurun_out/two_r0.out 2> gpurun_out/two_r0.err
wait
echo "rank0 stdout:"; cat gpurun_out/two_r0.out | cut -c1-400
echo "rank1 stdout bytes: $(wc -c < gpurun_out/two_r1.out)"
tail -2 gpurun_out/two_r0.err | cut -c1-300; tail -2 gpurun_out/two_r1.err | cut -c1-300
