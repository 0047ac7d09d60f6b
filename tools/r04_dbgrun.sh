cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=8
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "deferred or lane_compaction or phase_refill or tail_wait or batch_matches or leave_the_central or multi_phase or staged_and_plain" 2>&1 | tail -3
for rep in 1 2; do
MPC_TAIL_HOLD_BATCHES=20 timeout -k 5 100 python tools/hold_test.py --tail-cut 20 2>&1 | tail -1
MPC_TAIL_HOLD_BATCHES=20 timeout -k 5 100 python tools/hold_test.py --tail-cut 20 --population filtered 2>&1 | tail -1
done
for k in 1 2; do timeout -k 10 200 python bench.py --no-legs --no-cpu-baseline --no-host-leg --steps 200 --full-json /tmp/x.json > /dev/null 2>&1; python -c "
import json; r=json.load(open('/tmp/x.json')); print('headline survey K=200: %.2f M  kernel_ms %.3f' % (r['value']/1e6, r['roofline']['kernel_ms_avg']))"; done
