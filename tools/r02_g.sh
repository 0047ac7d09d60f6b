#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r02g_pytest.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r02g_pytest.log
python bench.py --steps 40 --no-cpu-baseline > gpurun_out/r02g_bench.json 2> gpurun_out/r02g.err; echo "bench rc=$?"
python tools/show_bench.py gpurun_out/r02g_bench.json
MPC_LDS=0 python bench.py --steps 40 --no-cpu-baseline > gpurun_out/r02g_bench_nolds.json 2>> gpurun_out/r02g.err
python tools/show_bench.py gpurun_out/r02g_bench_nolds.json | grep "host path"
BS=1,64,256,1024,2048,4096,8192,16384,65536 python tools/batch_sweep.py > gpurun_out/r02g_batch_sweep.txt 2>&1; BS=1,64,256,1024,2048,4096,8192,16384,65536 MPC_LDS=0 python tools/batch_sweep.py > gpurun_out/r02g_batch_sweep_nolds.txt 2>&1
paste gpurun_out/r02g_batch_sweep.txt gpurun_out/r02g_batch_sweep_nolds.txt | head -20
for q in 8 16; do for fl in 8 16; do
  GPU_MAX_HW_QUEUES=$q python bench.py --precision f32 --weights-sweep --no-traj --inflight $fl --steps 64 --warmup 16 --no-cpu-baseline --no-host-leg > gpurun_out/r02g_f32_sweep_q${q}_f$fl.json 2>> gpurun_out/r02g.err; echo "f32 sweep hwq=$q inflight=$fl rc=$?"
  python tools/show_bench.py gpurun_out/r02g_f32_sweep_q${q}_f$fl.json | head -1
done; done
