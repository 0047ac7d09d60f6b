#!/bin/bash
# extra measurements: other BASELINE.json configs at 1 GPU, and a 2-rank rehearsal of the multi-process bench path
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; TAG=${1:-extra}
cd $R
echo "== configs[1]: 4096 straight... (bench uses lake generator; straight-line batch is covered by tests) B=4096 lake" 
timeout -k 10 300 python bench.py --batch 4096 --no-cpu-baseline > $OUT/${TAG}_b4096.json 2> $OUT/${TAG}_b4096.err; echo "exit=$?"; cat $OUT/${TAG}_b4096.json | cut -c1-300
echo "== configs[3] share of one GPU: 32768 x N=25 dt=0.05 config-stable"
timeout -k 10 300 python bench.py --batch 32768 --N 25 --dt 0.05 --config config-stable.json --steps 10 --cpu-seconds 6 > $OUT/${TAG}_n25.json 2> $OUT/${TAG}_n25.err; echo "exit=$?"; cat $OUT/${TAG}_n25.json | cut -c1-400
echo "== configs[4] in fp64: 131072 x N=10 weight sweep"
timeout -k 10 300 python bench.py --batch 131072 --weights-sweep --no-traj --steps 10 --cpu-seconds 6 > $OUT/${TAG}_w.json 2> $OUT/${TAG}_w.err; echo "exit=$?"; cat $OUT/${TAG}_w.json | cut -c1-400
echo "== 2-rank rehearsal on one GPU (gloo, both ranks on cuda:0)"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 --batch 16384 --backend gloo --single-device > $OUT/${TAG}_2rank.json 2> $OUT/${TAG}_2rank.err; echo "exit=$?"; tail -1 $OUT/${TAG}_2rank.json | cut -c1-400; tail -3 $OUT/${TAG}_2rank.err
