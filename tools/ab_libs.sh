#!/bin/bash
# same-box A/B of several builds of libmpc_amd.so (gpurun_in/libmpc_<tag>.so): interleaved runs of tools/batch_sweep.py
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp carnd-mpc-project_amd/lib/libmpc_amd.so /tmp/libmpc_keep.so
for round in 1 2 3; do
  for f in gpurun_in/libmpc_*.so; do
    cp $f carnd-mpc-project_amd/lib/libmpc_amd.so
    echo "== round $round $(basename $f)"
    BS=${BS:-1024,65536} REPS=${REPS:-20} NOSYNC=${NOSYNC:-} timeout -k 10 120 python tools/batch_sweep.py 2>&1 | grep "^B"
  done
done
cp /tmp/libmpc_keep.so carnd-mpc-project_amd/lib/libmpc_amd.so
