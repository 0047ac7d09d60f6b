#!/bin/bash
# collective-path rehearsals that one GPU allows: RCCL with a single rank; two self-spawned ranks sharing the GPU over gloo
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r02i_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02i_pytest.log
python bench.py --force-collective --steps 60 --no-cpu-baseline --no-host-leg > gpurun_out/r02i_rccl1.json 2> gpurun_out/r02i.err; echo "rccl single-rank async rc=$?"
python tools/show_bench.py gpurun_out/r02i_rccl1.json | head -1; python -c "import json; r=json.load(open('gpurun_out/r02i_rccl1.json')); print('   ', r['config']['collective_mode'], r['config']['gather_checked'])"
python bench.py --force-collective --no-overlap --steps 60 --no-cpu-baseline --no-host-leg > gpurun_out/r02i_rccl1_sync.json 2>> gpurun_out/r02i.err; echo "rccl single-rank sync rc=$?"
python tools/show_bench.py gpurun_out/r02i_rccl1_sync.json | head -1; python -c "import json; r=json.load(open('gpurun_out/r02i_rccl1_sync.json')); print('   ', r['config']['collective_mode'], r['config']['gather_checked'])"
python bench.py --gpus 2 --single-device --backend gloo --batch 32768 --steps 30 --no-cpu-baseline --no-host-leg > gpurun_out/r02i_spawn2.json 2>> gpurun_out/r02i.err; echo "self-spawn 2 ranks on one GPU (gloo) rc=$?"
python tools/show_bench.py gpurun_out/r02i_spawn2.json | head -1; python -c "import json; r=json.load(open('gpurun_out/r02i_spawn2.json')); print('   ', r['n_gpus'], r['config']['collective_mode'], r['config']['gather_checked'])"
tail -3 gpurun_out/r02i.err
