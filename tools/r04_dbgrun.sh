# same-box A/B: staging loads non-temporal (gpurun_in/libmpc_amd_nt.so) against the shipped build, N = 25 share and the headline
cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=8
cp carnd-mpc-project_amd/lib/libmpc_amd.so /tmp/libmpc_keep.so
one() { timeout -k 10 200 python bench.py --no-legs --no-cpu-baseline --no-host-leg --full-json /tmp/ab.json "$@" > /dev/null 2>/tmp/ab.err; python -c "
import json; r=json.load(open('/tmp/ab.json')); print('   %.3f M solves/s  kernel_ms %.3f' % (r['value']/1e6, r['roofline']['kernel_ms_avg']))"; }
for round in 1 2; do
  for lib in plain nt; do
    if [ $lib = nt ]; then cp gpurun_in/libmpc_amd_nt.so carnd-mpc-project_amd/lib/libmpc_amd.so; else cp /tmp/libmpc_keep.so carnd-mpc-project_amd/lib/libmpc_amd.so; fi
    echo "== round $round $lib: N=25 share (survey), 150 steps"; one --steps 150 --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4
    echo "== round $round $lib: headline (survey), 200 steps"; one --steps 200
  done
done
cp /tmp/libmpc_keep.so carnd-mpc-project_amd/lib/libmpc_amd.so
