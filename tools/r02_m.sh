#!/bin/bash
# does the footprint of the workspaces in flight matter (Infinity Cache 256 MB)?  same concurrency, different footprints
set -o pipefail
mkdir -p gpurun_out
b() { local name=$1; shift; python bench.py "$@" --no-cpu-baseline --no-host-leg > gpurun_out/r02m_$name.json 2>> gpurun_out/r02m.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r02m_$name.json | head -1; }
b b65536_f2 --steps 60
b b32768_f2 --batch 32768 --inflight 2 --steps 120
b b32768_f3 --batch 32768 --inflight 3 --steps 120
b b32768_f4 --batch 32768 --inflight 4 --steps 120
b b16384_f4 --batch 16384 --inflight 4 --steps 240
b b49152_f2 --batch 49152 --inflight 2 --steps 80
b f32_b65536_f4 --precision f32 --inflight 4 --steps 60
b f32_b32768_f4 --precision f32 --batch 32768 --inflight 4 --steps 120
b f32_b32768_f8 --precision f32 --batch 32768 --inflight 8 --steps 120
