#!/bin/bash
# final session of round 2: everything the driver will run, plus the profiles that DESIGN.md quotes
set -o pipefail
mkdir -p gpurun_out
bash tools/gpu_round.sh r02 > gpurun_out/r02_round.log 2>&1; echo "round f64 rc=$?"
SKIP_TESTS=1 BENCH_ARGS="--precision f32 --inflight 4" bash tools/gpu_round.sh r02f32 > gpurun_out/r02f32_round.log 2>&1; echo "round f32 rc=$?"
tail -3 gpurun_out/r02_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()"
python tools/show_bench.py gpurun_out/r02_bench.json gpurun_out/r02f32_bench.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 30 --warmup 4 --no-cpu-baseline --no-host-leg > gpurun_out/r02_torchrun1.json 2> gpurun_out/r02_torchrun1.err; echo "torchrun nproc 1 rc=$?"; python tools/show_bench.py gpurun_out/r02_torchrun1.json | head -1
python bench.py --precision f32 --weights-sweep --no-traj --inflight 8 --steps 64 --warmup 8 --no-host-leg > gpurun_out/r02_cfg4_per_gpu.json 2>> gpurun_out/r02final.err; python tools/show_bench.py gpurun_out/r02_cfg4_per_gpu.json
python bench.py --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4 --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/r02_cfg3_per_gpu.json 2>> gpurun_out/r02final.err; python tools/show_bench.py gpurun_out/r02_cfg3_per_gpu.json
python bench.py --batch 4096 --config config-stable.json --inflight 1 --steps 64 --no-cpu-baseline > gpurun_out/r02_cfg1.json 2>> gpurun_out/r02final.err; python tools/show_bench.py gpurun_out/r02_cfg1.json
python bench.py --unfiltered --steps 40 --no-cpu-baseline --no-host-leg > gpurun_out/r02_unfiltered.json 2>> gpurun_out/r02final.err; python tools/show_bench.py gpurun_out/r02_unfiltered.json
