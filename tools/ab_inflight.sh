#!/bin/bash
# same-box A/B of builds of libmpc_amd.so (gpurun_in/libmpc_<tag>.so) in the steady state (tools/inflight_bench.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp carnd-mpc-project_amd/lib/libmpc_amd.so /tmp/libmpc_keep.so
for round in 1 2; do
  for f in gpurun_in/libmpc_*.so; do
    cp $f carnd-mpc-project_amd/lib/libmpc_amd.so
    echo "== round $round $(basename $f)"
    timeout -k 10 120 python tools/inflight_bench.py 2>&1 | grep "in flight" | cut -c1-60
  done
done
cp /tmp/libmpc_keep.so carnd-mpc-project_amd/lib/libmpc_amd.so
