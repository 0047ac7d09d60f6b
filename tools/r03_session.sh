#!/bin/bash
# The GPU sessions of round 3 (one parameterised script; run from the repo root through gpurun):
#   bash tools/r03_session.sh a     parity tests + bench line + same-box A/B of the builds in gpurun_in/
set -o pipefail
S=${1:-a}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
P=$OUT/r03${S}_progress.log
echo "== start $S" | tee $P
case $S in
a)
  timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/r03a_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -5 $OUT/r03a_pytest.log
  timeout -k 10 400 python bench.py --steps 40 > $OUT/r03a_bench.json 2> $OUT/r03a_bench.err; echo "bench exit=$?" | tee -a $P; python tools/show_bench.py $OUT/r03a_bench.json | tee -a $P
  bash tools/ab_inflight.sh 2>&1 | tee $OUT/r03a_ab_inflight.log
  ;;
t)   # parity tests only
  timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/r03t_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -15 $OUT/r03t_pytest.log
  ;;
k)   # selected tests: K="expr"
  timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "$K" > $OUT/r03k_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -25 $OUT/r03k_pytest.log
  ;;
d)   # deferred tails: cut sweep on every workload (no legs, no CPU / host legs)
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" > $OUT/r03d_$tag.json 2> $OUT/r03d_$tag.err; echo "$tag exit=$?" | tee -a $P; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03d_$tag.json"))
    print("   %-22s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s  tails %s" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["mean_iterations"], r["max_iterations"], {k: v for k, v in r["status_counts"].items() if v}, r["config"]["deferred_tails"] if isinstance(r["config"]["deferred_tails"], str) else {k: r["config"]["deferred_tails"][k] for k in ("tail_launches", "instances_over_the_cut_in_the_last_batch")}))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  for c in 0 10 12 14 16 20; do run head_c$c --steps 60 --tail-cut $c; done
  for c in 0 12 16 20 30; do run survey_c$c --steps 30 --population survey --tail-cut $c; done
  MPC_TAIL_PRIORITY=low run survey_c16_lowprio --steps 30 --population survey --tail-cut 16
  MPC_TAIL_PRIORITY=normal run survey_c16_normprio --steps 30 --population survey --tail-cut 16
  for c in 0 16 24 32 48; do run w64_c$c --steps 30 --weights-sweep --inflight 4 --tail-cut $c; done
  for c in 0 16 24 32 48; do run w32_c$c --steps 30 --weights-sweep --precision f32 --no-traj --batch 131072 --inflight 4 --tail-cut $c; done
  for c in 0 16 24 32; do run n25_c$c --steps 20 --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4 --tail-cut $c; done
  ;;
e)   # deferred tails: steady state (more steps), capacity, tail waves; kernel trace of one configuration
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" > $OUT/r03e_$tag.json 2> $OUT/r03e_$tag.err; echo "$tag exit=$?" | tee -a $P; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03e_$tag.json"))
    print("   %-22s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s  tails %s" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["mean_iterations"], r["max_iterations"], {k: v for k, v in r["status_counts"].items() if v}, r["config"]["deferred_tails"] if isinstance(r["config"]["deferred_tails"], str) else {k: r["config"]["deferred_tails"][k] for k in ("tail_launches", "instances_over_the_cut_in_the_last_batch")}))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  run survey_c16_s200 --steps 200 --population survey --tail-cut 16
  run survey_c20_s200 --steps 200 --population survey --tail-cut 20
  run survey_c20_s400 --steps 400 --population survey --tail-cut 20
  run head_c20_s200 --steps 200 --tail-cut 20
  run head_c0_s200 --steps 200 --tail-cut 0
  run w32_c24_s100 --steps 100 --weights-sweep --precision f32 --no-traj --batch 131072 --inflight 4 --tail-cut 24
  run w32_c24_s100_i8 --steps 100 --weights-sweep --precision f32 --no-traj --batch 131072 --inflight 8 --tail-cut 24
  MPC_TAIL_WAVES=512 run w32_c24_s100_w512 --steps 100 --weights-sweep --precision f32 --no-traj --batch 131072 --inflight 4 --tail-cut 24
  MPC_TAIL_WAVES=128 run w32_c24_s100_w128 --steps 100 --weights-sweep --precision f32 --no-traj --batch 131072 --inflight 4 --tail-cut 24
  run w64_c24_s100 --steps 100 --weights-sweep --inflight 4 --tail-cut 24
  run n25_c24_s60 --steps 60 --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4 --tail-cut 24
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03e_prof_w32 -o trace -- python3 $R/bench.py --steps 40 --warmup 2 --no-legs --no-cpu-baseline --no-host-leg --weights-sweep --precision f32 --no-traj --batch 131072 --inflight 4 --tail-cut 24 > $OUT/r03e_prof_w32.json 2> $OUT/r03e_prof_w32.err; echo "rocprof exit=$?" | tee -a $P
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03e_prof_survey -o trace -- python3 $R/bench.py --steps 60 --warmup 2 --no-legs --no-cpu-baseline --no-host-leg --population survey --tail-cut 20 > $OUT/r03e_prof_survey.json 2> $OUT/r03e_prof_survey.err; echo "rocprof exit=$?" | tee -a $P
  cd $R
  ;;
f)   # deferred tails with several tail streams and a longer ring
  timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "deferred" > $OUT/r03f_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -5 $OUT/r03f_pytest.log
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" > $OUT/r03f_$tag.json 2> $OUT/r03f_$tag.err; echo "$tag exit=$?" | tee -a $P; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03f_$tag.json"))
    print("   %-22s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s  tails %s" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["mean_iterations"], r["max_iterations"], {k: v for k, v in r["status_counts"].items() if v}, r["config"]["deferred_tails"] if isinstance(r["config"]["deferred_tails"], str) else {k: r["config"]["deferred_tails"][k] for k in ("tail_launches", "instances_over_the_cut_in_the_last_batch")}))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  for st in 1 2 3; do for ring in 32 64; do MPC_TAIL_STREAMS=$st run survey_c20_st${st}_r$ring --steps 300 --population survey --tail-cut 20 --tail-ring $ring; done; done
  MPC_TAIL_STREAMS=2 run survey_c16_st2_r64 --steps 300 --population survey --tail-cut 16 --tail-ring 64
  MPC_TAIL_STREAMS=2 run head_c20_st2 --steps 200 --tail-cut 20
  MPC_TAIL_STREAMS=2 run w32_c24_st2 --steps 100 --weights-sweep --precision f32 --no-traj --batch 131072 --inflight 4 --tail-cut 24
  MPC_TAIL_STREAMS=2 run w32_c20_st2 --steps 100 --weights-sweep --precision f32 --no-traj --batch 131072 --inflight 4 --tail-cut 20
  MPC_TAIL_STREAMS=2 run w64_c24_st2 --steps 100 --weights-sweep --inflight 4 --tail-cut 24
  MPC_TAIL_STREAMS=2 run n25_c24_st2 --steps 60 --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4 --tail-cut 24
  ;;
b)   # the default bench line as the driver runs it
  ( time timeout -k 10 900 python bench.py > $OUT/r03b_bench.json 2> $OUT/r03b_bench.err ) 2>&1 | tail -3 | tee -a $P; python tools/show_bench.py $OUT/r03b_bench.json | tee -a $P
  ;;
m)   # mixed precision: tests, then rates (fp32 handle: mixed vs pure; fp64 handle: fp32 start vs plain)
  timeout -k 10 900 python -m pytest tests -m gpu -q -k "f32 or tile_pool or multi_phase or lds_resident or deferred or config3" > $OUT/r03m_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -8 $OUT/r03m_pytest.log
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" > $OUT/r03m_$tag.json 2> $OUT/r03m_$tag.err; echo "$tag exit=$?" | tee -a $P; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03m_$tag.json"))
    print("   %-22s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s  kernel_ms %.3f" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["mean_iterations"], r["max_iterations"], {k: v for k, v in r["status_counts"].items() if v}, r["roofline"]["kernel_ms_avg"]))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  for i in 2 4; do run head_f64_i$i --steps 60 --inflight $i; done
  for i in 2 3 4; do run head_f64_f32start_i$i --steps 60 --inflight $i --f64-f32-start; done
  run head_f64_f32start_mu1e3 --steps 60 --inflight 3 --f64-f32-start --switch-mu 1e-3
  for i in 2 4; do run head_f32_mixed_i$i --steps 60 --precision f32 --inflight $i; done
  run head_f32_pure_i4 --steps 60 --precision f32 --f32-pure --inflight 4
  for i in 4 8; do run w32_mixed_i$i --steps 40 --weights-sweep --precision f32 --no-traj --batch 131072 --inflight $i; done
  run w32_pure_i4 --steps 40 --weights-sweep --precision f32 --f32-pure --no-traj --batch 131072 --inflight 4
  run w32_pure_i4_c24 --steps 60 --weights-sweep --precision f32 --f32-pure --no-traj --batch 131072 --inflight 4 --tail-cut 24
  ;;
n)   # mixed precision with deferred tails; switch_mu sweep on the fp64 headline
  timeout -k 10 900 python -m pytest tests -m gpu -q -k "deferred or f32" > $OUT/r03n_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -8 $OUT/r03n_pytest.log
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" > $OUT/r03n_$tag.json 2> $OUT/r03n_$tag.err; echo "$tag exit=$?" | tee -a $P; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03n_$tag.json"))
    print("   %-22s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s  kernel_ms %.3f" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["mean_iterations"], r["max_iterations"], {k: v for k, v in r["status_counts"].items() if v}, r["roofline"]["kernel_ms_avg"]))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  for c in 8 12 16 24; do run w32_mixed_c$c --steps 80 --weights-sweep --precision f32 --no-traj --batch 131072 --inflight 4 --tail-cut $c; done
  run w32_mixed_c12_i8 --steps 80 --weights-sweep --precision f32 --no-traj --batch 131072 --inflight 8 --tail-cut 12
  for mu in 2e-6 2e-5 2e-4; do run head_f64_f32start_mu$mu --steps 100 --inflight 3 --f64-f32-start --switch-mu $mu; done
  run head_f64_plain --steps 100 --inflight 2
  run head_f64_f32start_i3_b --steps 100 --inflight 3 --f64-f32-start
  run head_f64_plain_b --steps 100 --inflight 2
  run head_f64_f32start_c10 --steps 200 --inflight 3 --f64-f32-start --tail-cut 10
  run survey_f32start_c12 --steps 300 --inflight 3 --f64-f32-start --tail-cut 12 --population survey --tail-ring 64
  run w64_f32start_c12 --steps 80 --weights-sweep --inflight 4 --f64-f32-start --tail-cut 12
  run n25_f32start_c12 --steps 60 --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4 --f64-f32-start --tail-cut 12
  ;;
p)   # parity of the fp64 solve with its early iterations on the fp32 record (MPC_MIXED=1 forces f64_f32_start on every fp64 handle)
  MPC_MIXED=1 timeout -k 10 900 python -m pytest tests -m gpu -q -k "matches_oracle or scipy or full_size_properties or soak or test_cpp or horizon_extremes or rollout or plot_anchors or run_path" > $OUT/r03p_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -12 $OUT/r03p_pytest.log
  ;;
g)   # profiles: the default headline, then the same with its early iterations on the fp32 record (A/B on one box)
  SKIP_TESTS=1 bash tools/gpu_round.sh r03
  SKIP_TESTS=1 BENCH_ARGS="--f64-f32-start --inflight 4" bash tools/gpu_round.sh r03fs
  ;;
s)   # the soak tests at full size (every instance against the oracle on the box's cores)
  MPC_SOAK=1 timeout -k 10 1100 python -m pytest tests/test_soak.py -m gpu -q -s > $OUT/r03s_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -5 $OUT/r03s_pytest.log
  ;;
c)   # collective rehearsal on one GPU: RCCL initialised with a single rank, the collectives inside the timed region
  timeout -k 10 300 python bench.py --steps 80 --no-legs --no-cpu-baseline --no-host-leg > $OUT/r03c_none.json 2> $OUT/r03c_none.err
  python -c "import json; r=json.load(open('$OUT/r03c_none.json')); print('   no collective', r['value']/1e6, 'M solves/s')" | tee -a $P
  for g in root all; do for grp in 1 4 8; do for extra in "" "--gather-results-only"; do
    tag=${g}_g${grp}${extra:+_ro}
    timeout -k 10 300 python bench.py --force-collective --gather $g --gather-group $grp $extra --steps 80 --no-legs --no-cpu-baseline --no-host-leg > $OUT/r03c_$tag.json 2> $OUT/r03c_$tag.err; echo "$tag exit=$?" | tee -a $P
    python -c "import json; r=json.load(open('$OUT/r03c_$tag.json')); print('  ', r['value']/1e6, 'M solves/s', r['config']['collective_mode'], 'checked', r['config']['gather_checked'], 'bytes/rank/batch', r['config']['gather_bytes_sent_per_rank_per_batch'])" | tee -a $P
  done; done; done
  timeout -k 10 300 python bench.py --force-collective --population survey --tail-cut 20 --tail-ring 64 --steps 200 --no-legs --no-cpu-baseline --no-host-leg > $OUT/r03c_tails.json 2> $OUT/r03c_tails.err; echo "tails exit=$?" | tee -a $P
  python -c "import json; r=json.load(open('$OUT/r03c_tails.json')); print('  ', r['value']/1e6, 'M solves/s', r['config']['collective_mode'], 'checked', r['config']['gather_checked'], r['status_counts'])" | tee -a $P
  timeout -k 10 300 python bench.py --gpus 2 --single-device --backend gloo --steps 20 --no-legs --no-cpu-baseline --no-host-leg > $OUT/r03c_gloo2.json 2> $OUT/r03c_gloo2.err; echo "gloo2 exit=$?" | tee -a $P
  python -c "import json; r=json.load(open('$OUT/r03c_gloo2.json')); print('  ', r['value']/1e6, 'M solves/s', r['n_gpus'], r['config']['collective_mode'], 'checked', r['config']['gather_checked'])" | tee -a $P
  ;;
q)   # the long-horizon share and the fp64 weight sweep: plain / fp32 start, with and without deferred tails
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" > $OUT/r03q_$tag.json 2> $OUT/r03q_$tag.err; echo "$tag exit=$?" | tee -a $P; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03q_$tag.json"))
    print("   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["mean_iterations"], r["max_iterations"], {k: v for k, v in r["status_counts"].items() if v}))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  N25="--N 25 --dt 0.05 --config config-stable.json --batch 32768 --steps 60"
  run n25_plain_i4 $N25 --inflight 4
  run n25_plain_i4_c24 $N25 --inflight 4 --tail-cut 24 --tail-ring 64
  run n25_f32s_i4 $N25 --inflight 4 --f64-f32-start
  run n25_f32s_i6 $N25 --inflight 6 --f64-f32-start
  run n25_f32s_i4_c16 $N25 --inflight 4 --f64-f32-start --tail-cut 16 --tail-ring 64
  run n25_f32s_i6_c24 $N25 --inflight 6 --f64-f32-start --tail-cut 24 --tail-ring 64
  W="--weights-sweep --steps 80"
  run w64_plain_i4_c24 $W --inflight 4 --tail-cut 24 --tail-ring 64
  run w64_f32s_i4 $W --inflight 4 --f64-f32-start
  run w64_f32s_i6_c16 $W --inflight 6 --f64-f32-start --tail-cut 16 --tail-ring 64
  run w64_f32s_i6_c24 $W --inflight 6 --f64-f32-start --tail-cut 24 --tail-ring 64
  run c1_plain --config config-stable.json --batch 4096 --steps 200 --inflight 2
  run c1_f32s --config config-stable.json --batch 4096 --steps 200 --inflight 3 --f64-f32-start
  run c1_plain_i4 --config config-stable.json --batch 4096 --steps 200 --inflight 4
  ;;
r)   # small batches want many in flight; the long-horizon share with more in flight
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" > $OUT/r03r_$tag.json 2> $OUT/r03r_$tag.err; echo "$tag exit=$?" | tee -a $P; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03r_$tag.json"))
    print("   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["mean_iterations"], r["max_iterations"], {k: v for k, v in r["status_counts"].items() if v}))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  C1="--config config-stable.json --batch 4096 --steps 400"
  for i in 6 8 12 16; do run c1_plain_i$i $C1 --inflight $i; done
  for i in 8 16; do run c1_f32s_i$i $C1 --inflight $i --f64-f32-start; done
  N25="--N 25 --dt 0.05 --config config-stable.json --batch 32768 --steps 60"
  run n25_f32s_i8 $N25 --inflight 8 --f64-f32-start
  run n25_plain_i8 $N25 --inflight 8
  run n25_plain_i8_c24 $N25 --inflight 8 --tail-cut 24 --tail-ring 64
  ;;
c2)  # where does the cost of a collective sit?  one gather for the whole run, and RCCL initialised but never used
  for grp in 40 80; do
    timeout -k 10 300 python bench.py --force-collective --gather root --gather-group $grp --steps 80 --no-legs --no-cpu-baseline --no-host-leg > $OUT/r03c2_g$grp.json 2> $OUT/r03c2_g$grp.err; echo "g$grp exit=$?" | tee -a $P
    python -c "import json; r=json.load(open('$OUT/r03c2_g$grp.json')); print('  ', r['value']/1e6, 'M solves/s', r['config']['collective_mode'], 'checked', r['config']['gather_checked'])" | tee -a $P
  done
  MPC_BENCH_NO_COLLECTIVE_CALLS=1 timeout -k 10 300 python bench.py --force-collective --gather root --steps 80 --no-legs --no-cpu-baseline --no-host-leg > $OUT/r03c2_initonly.json 2> $OUT/r03c2_initonly.err; echo "initonly exit=$?" | tee -a $P
  python -c "import json; r=json.load(open('$OUT/r03c2_initonly.json')); print('   RCCL initialised, no collective issued:', r['value']/1e6, 'M solves/s')" | tee -a $P
  timeout -k 10 300 python bench.py --steps 80 --no-legs --no-cpu-baseline --no-host-leg > $OUT/r03c2_none.json 2> $OUT/r03c2_none.err
  python -c "import json; r=json.load(open('$OUT/r03c2_none.json')); print('   no RCCL:', r['value']/1e6, 'M solves/s')" | tee -a $P
  timeout -k 10 300 python bench.py --force-collective --population survey --tail-cut 20 --tail-ring 64 --steps 200 --no-legs --no-cpu-baseline --no-host-leg > $OUT/r03c_tails.json 2> $OUT/r03c_tails.err; echo "tails exit=$?" | tee -a $P
  python -c "import json; r=json.load(open('$OUT/r03c_tails.json')); print('  ', r['value']/1e6, 'M solves/s', r['config']['collective_mode'], 'checked', r['config']['gather_checked'], r['status_counts'])" | tee -a $P
  ;;
c3)  # RCCL's kernels take SIMDs from the solve while they run: fewer channels
  for ch in default 1 2 4; do for grp in 1 4; do
    if [ $ch = default ]; then E=""; else E="NCCL_MAX_NCHANNELS=$ch NCCL_MIN_NCHANNELS=1"; fi
    env $E timeout -k 10 300 python bench.py --force-collective --gather root --gather-group $grp --steps 80 --no-legs --no-cpu-baseline --no-host-leg > $OUT/r03c3_ch${ch}_g$grp.json 2> $OUT/r03c3_ch${ch}_g$grp.err; echo "ch$ch g$grp exit=$?" | tee -a $P
    python -c "import json; r=json.load(open('$OUT/r03c3_ch${ch}_g$grp.json')); print('  ', r['value']/1e6, 'M solves/s', r['config']['collective_mode'], 'checked', r['config']['gather_checked'])" | tee -a $P
  done; done
  ;;
u)   # the long-horizon legs: does the step count explain the gap between the default run's legs and the standalone runs?
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" > $OUT/r03u_$tag.json 2> $OUT/r03u_$tag.err; echo "$tag exit=$?" | tee -a $P; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03u_$tag.json"))
    print("   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["mean_iterations"], r["max_iterations"]))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  N25="--N 25 --dt 0.05 --config config-stable.json --batch 32768"
  for st in 40 80 160; do run n25_f32s_i8_s$st $N25 --inflight 8 --f64-f32-start --steps $st; done
  for st in 40 80 160; do run n25_plain_i4_c24_s$st $N25 --inflight 4 --tail-cut 24 --tail-ring 64 --steps $st; done
  ;;
v)   # one leg alone, with and without HIP events around its launches
  for k in 1 2; do
    python bench.py --leg headline_f32_start 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   events   ', r['solves_per_s']/1e6, r.get('kernel_ms_avg'))" | tee -a $P
    MPC_BENCH_LEG_NO_EVENTS=1 python bench.py --leg headline_f32_start 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   no events', r['solves_per_s']/1e6)" | tee -a $P
    python bench.py --f64-f32-start --inflight 3 --steps 60 --no-legs --no-cpu-baseline --no-host-leg 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   main path', r['value']/1e6)" | tee -a $P
  done
  ;;
w)   # main path vs leg path of the same workload: which difference matters?
  m() { python bench.py --f64-f32-start --inflight 3 --steps 60 --no-legs --no-cpu-baseline --no-host-leg "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   main', '$*', r['value']/1e6, 'queues', r['config']['hw_queues'])" | tee -a $P; }
  m
  GPU_MAX_HW_QUEUES=4 m
  GPU_MAX_HW_QUEUES=8 m
  m --warmup 2
  m --tail-ring 64
  python bench.py --leg headline_f32_start 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   leg (queues 4 set)', r['solves_per_s']/1e6)" | tee -a $P
  ;;
x)   # configs[3] and configs[4]: the FULL batch on one GPU
  run() { tag=$1; shift; timeout -k 10 600 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" > $OUT/r03x_$tag.json 2> $OUT/r03x_$tag.err; echo "$tag exit=$?" | tee -a $P; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03x_$tag.json"))
    print("   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["mean_iterations"], r["max_iterations"], {k: v for k, v in r["status_counts"].items() if v}))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  C3="--N 25 --dt 0.05 --config config-stable.json --batch 262144 --steps 12 --warmup 4"
  run cfg3_full_plain $C3 --inflight 2
  run cfg3_full_tails $C3 --inflight 2 --tail-cut 24 --tail-ring 16
  run cfg3_full_f32start $C3 --inflight 3 --f64-f32-start
  C4="--weights-sweep --precision f32 --no-traj --batch 1048576 --steps 12 --warmup 4"
  run cfg4_full_mixed $C4 --inflight 2
  run cfg4_full_pure $C4 --inflight 2 --f32-pure
  run cfg4_full_pure_tails $C4 --inflight 2 --f32-pure --tail-cut 24 --tail-ring 16
  ;;
y)   # does lane refill pay on the fp32 kernel (less traffic-bound than the fp64 one)?
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" > $OUT/r03y_$tag.json 2> $OUT/r03y_$tag.err; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03y_$tag.json"))
    print("   %-26s %8.3f M solves/s  %.3f ms/batch  kernel_ms %.3f alone %.3f" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["roofline"]["kernel_ms_avg"], r["roofline"]["kernel_ms_alone"]))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  F="--precision f32 --f32-pure --steps 100"
  for ipl in 1 2 4; do for i in 2 4 6; do MPC_INSTANCES_PER_LANE=$ipl run f32pure_ipl${ipl}_i$i $F --inflight $i; done; done
  for ipl in 2 4; do MPC_INSTANCES_PER_LANE=$ipl MPC_REFILL_MIN=4 MPC_REFILL_WAIT=2 run f32pure_ipl${ipl}_i4_eager $F --inflight 4; done
  ;;
z)   # waves per tail launch on the SURVEY population (few stragglers per batch, 28 of them with very long chains)
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" > $OUT/r03z_$tag.json 2> $OUT/r03z_$tag.err; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03z_$tag.json"))
    print("   %-26s %8.3f M solves/s  %.3f ms/batch  kernel_ms %.3f  tails %s" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["roofline"]["kernel_ms_avg"], r["config"]["deferred_tails"]["tail_launches"]))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  S="--population survey --tail-cut 20 --tail-ring 64 --steps 500"
  for w in 8 16 32 64 128 256; do MPC_TAIL_WAVES=$w run survey_w$w $S; done
  MPC_TAIL_WAVES=32 run survey_w32_c16 --population survey --tail-cut 16 --tail-ring 64 --steps 500
  MPC_TAIL_WAVES=32 MPC_TAIL_STREAMS=3 run survey_w32_st3 $S
  ;;
h)   # where should the fp32 phase of the fp64 headline hand over?  (tol_f32 = the E_0 at which it does, switch_mu = the barrier value)
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --f64-f32-start --inflight 3 --steps 100 "$@" > $OUT/r03h_$tag.json 2> $OUT/r03h_$tag.err; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03h_$tag.json"))
    print("   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["mean_iterations"], r["max_iterations"], {k: v for k, v in r["status_counts"].items() if v}))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  run default
  run tol1e-3 --tol-f32 1e-3
  run tol2e-4 --tol-f32 2e-4
  run tol1e-4 --tol-f32 1e-4
  run tol1e-4_mu2e-6 --tol-f32 1e-4 --switch-mu 2e-6
  run tol5e-5_mu2e-6 --tol-f32 5e-5 --switch-mu 2e-6
  run tol2e-4_mu5e-6 --tol-f32 2e-4 --switch-mu 5e-6
  run default_again
  timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --steps 100 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   plain fp64               ', r['value']/1e6)" | tee -a $P
  ;;
fd)   # the fp64 phase of a mixed solve as a narrower grid whose lanes take the promoted instances in turn
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --f64-f32-start --steps 100 "$@" > $OUT/r03fd_$tag.json 2> $OUT/r03fd_$tag.err; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03fd_$tag.json"))
    print("   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["mean_iterations"], r["max_iterations"], {k: v for k, v in r["status_counts"].items() if v}))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  run div1_i3 --inflight 3
  for d in 2 3 4 6; do MPC_FINISH_DIV=$d run div${d}_i3 --inflight 3; done
  for d in 2 4; do MPC_FINISH_DIV=$d MPC_FINISH_REFILL_MIN=8 MPC_FINISH_REFILL_WAIT=4 run div${d}_i3_r8_4 --inflight 3; done
  for d in 2 4; do MPC_FINISH_DIV=$d MPC_FINISH_REFILL_MIN=1 MPC_FINISH_REFILL_WAIT=0 run div${d}_i3_r1_0 --inflight 3; done
  for d in 2 4; do MPC_FINISH_DIV=$d run div${d}_i4 --inflight 4; done
  for d in 2 4; do MPC_FINISH_DIV=$d run div${d}_i2 --inflight 2; done
  run div1_i3_again --inflight 3
  timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --steps 100 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   plain fp64               ', r['value']/1e6)" | tee -a $P
  ;;
lc)   # lane compaction (MPC_LANE_COMPACT=gap): bitwise test, then A/B on the headline, plain and fp32 start
  timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "lane_compaction" > $OUT/r03lc_pytest.log 2>&1; rc=$?; echo "pytest exit=$rc" | tee -a $P; tail -15 $OUT/r03lc_pytest.log
  if [ $rc -ne 0 ]; then exit 1; fi
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --steps 100 "$@" > $OUT/r03lc_$tag.json 2> $OUT/r03lc_$tag.err; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03lc_$tag.json"))
    print("   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s" % ("$tag", r["value"] / 1e6, r["ms_per_step"], r["mean_iterations"], r["max_iterations"], {k: v for k, v in r["status_counts"].items() if v}))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  for rep in a b; do
    for g in 0 1 2 3; do MPC_LANE_COMPACT=$g run plain_g${g}_$rep; done
    for g in 0 1 2 3; do MPC_LANE_COMPACT=$g run f32s_g${g}_$rep --f64-f32-start --inflight 3; done
  done
  for g in 0 2; do MPC_LANE_COMPACT=$g run plain_i3_g${g} --inflight 3; done
  ;;
lp)   # lane compaction: what the counters say it saves (FETCH_SIZE / WRITE_SIZE / SQ passes with and without)
  cd /tmp && export TMPDIR=/tmp
  for g in 0 2; do
    export MPC_LANE_COMPACT=$g
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/r03lp_g${g}_$c -o pmc -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-legs > /dev/null 2> $OUT/r03lp_g${g}_$c.err; echo "g=$g $c exit=$?" | tee -a $P
    done
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/r03lp_g${g}_SQ -o pmc -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-legs > /dev/null 2> $OUT/r03lp_g${g}_SQ.err; echo "g=$g SQ exit=$?" | tee -a $P
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03lp_g${g}_trace -o trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg --no-legs > /dev/null 2> $OUT/r03lp_g${g}_trace.err; echo "g=$g trace exit=$?" | tee -a $P
  done
  unset MPC_LANE_COMPACT
  cd $R
  python tools/pmc_mean.py $OUT/r03lp_g0_FETCH_SIZE $OUT/r03lp_g0_WRITE_SIZE $OUT/r03lp_g0_SQ $OUT/r03lp_g2_FETCH_SIZE $OUT/r03lp_g2_WRITE_SIZE $OUT/r03lp_g2_SQ | tee $OUT/r03lp_pmc.jsonl
  find $OUT -path "*r03lp_g*_trace*" -name "*kernel_stats.csv" | while read f; do echo $f; head -3 "$f" | cut -c1-60,400-; done
  ;;
ld)   # lane compaction on every leg of the default bench run, A/B/A/B
  for rep in a b; do for g in 0 2; do
    MPC_LANE_COMPACT=$g timeout -k 10 400 python bench.py --no-cpu-baseline > $OUT/r03ld_g${g}_$rep.json 2> $OUT/r03ld_g${g}_$rep.err; echo "g=$g $rep exit=$?" | tee -a $P
    python tools/show_bench.py $OUT/r03ld_g${g}_$rep.json | tee -a $P
  done; done
  ;;
le)   # lane compaction: the gap, per workload (the legs of the default run, one process each)
  leg() { tag=$1; name=$2; timeout -k 10 300 python bench.py --leg $name > $OUT/r03le_$tag.json 2> $OUT/r03le_$tag.err; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03le_$tag.json"))
    print("   %-34s %8.3f M solves/s  %.3f ms/batch" % ("$tag", r["solves_per_s"] / 1e6, r["ms_per_batch"]))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  for rep in a b; do for g in 0 1 2 3 4; do
    for l in configs_3_share configs_3_share_f32_start unfiltered configs_4_share_pure_fp32; do MPC_LANE_COMPACT=$g leg ${l}_g${g}_$rep $l; done
  done; done
  ;;
lf)   # lane compaction: quick A/B of a build (bitwise test, headline and the N = 25 share with and without)
  timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "lane_compaction" > $OUT/r03lf_pytest.log 2>&1; rc=$?; echo "pytest exit=$rc" | tee -a $P; tail -5 $OUT/r03lf_pytest.log
  if [ $rc -ne 0 ]; then exit 1; fi
  for rep in a b c; do for g in 0 2; do
    MPC_LANE_COMPACT=$g timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --steps 100 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   headline g=$g $rep  %.3f M' % (r['value']/1e6))" | tee -a $P
    MPC_LANE_COMPACT=$g timeout -k 10 300 python bench.py --leg configs_3_share 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   configs_3_share g=$g $rep  %.3f M' % (r['solves_per_s']/1e6))" | tee -a $P
  done; done
  ;;
us)   # the unfiltered population with deferred tails: run length (the drain of the last tails is inside the clock) and the cut
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --population survey --tail-ring 64 "$@" > $OUT/r03us_$tag.json 2> $OUT/r03us_$tag.err; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r03us_$tag.json"))
    print("   %-22s %8.3f M solves/s  %.3f ms/batch  status %s" % ("$tag", r["value"] / 1e6, r["ms_per_step"], {k: v for k, v in r["status_counts"].items() if v}))
except Exception as e:
    print("   $tag: no result", e)
PY
  }
  for st in 500 1000 2000 4000; do run c20_s$st --tail-cut 20 --steps $st; done
  for c in 14 16 18 22; do run c${c}_s2000 --tail-cut $c --steps 2000; done
  run c18_s2000_i3 --tail-cut 18 --steps 2000 --inflight 3
  ;;
lg)   # lane compaction: gap x cooldown (passes without another move), headline and the N = 25 share
  timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "lane_compaction or multi_phase or deferred_tails_are_bitwise" > $OUT/r03lg_pytest.log 2>&1; rc=$?; echo "pytest exit=$rc" | tee -a $P; tail -5 $OUT/r03lg_pytest.log
  if [ $rc -ne 0 ]; then exit 1; fi
  for rep in a b; do for gc in "0 2" "2 2" "2 0" "1 0" "1 1" "3 0"; do
    set -- $gc
    MPC_LANE_COMPACT=$1 MPC_LANE_COMPACT_COOLDOWN=$2 timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --steps 100 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   headline gap=$1 cooldown=$2 $rep  %.3f M' % (r['value']/1e6))" | tee -a $P
    MPC_LANE_COMPACT=$1 MPC_LANE_COMPACT_COOLDOWN=$2 timeout -k 10 300 python bench.py --leg configs_3_share 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   configs_3_share gap=$1 cooldown=$2 $rep  %.3f M' % (r['solves_per_s']/1e6))" | tee -a $P
  done; done
  ;;
lh)   # same-box A/B of the builds in gpurun_in/ (libmpc_<tag>.so): headline and the N = 25 share, lane compaction off / on
  cp carnd-mpc-project_amd/lib/libmpc_amd.so /tmp/libmpc_keep.so
  for rep in a b c; do for f in gpurun_in/libmpc_*.so; do
    cp $f carnd-mpc-project_amd/lib/libmpc_amd.so; t=$(basename $f .so)
    for gc in "0 2" "2 2" "2 0" "1 0"; do
      set -- $gc
      MPC_LANE_COMPACT=$1 MPC_LANE_COMPACT_COOLDOWN=$2 timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --steps 100 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   $t headline gap=$1 cooldown=$2 $rep  %.3f M' % (r['value']/1e6))" | tee -a $P
      MPC_LANE_COMPACT=$1 MPC_LANE_COMPACT_COOLDOWN=$2 timeout -k 10 300 python bench.py --leg configs_3_share 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   $t configs_3_share gap=$1 cooldown=$2 $rep  %.3f M' % (r['solves_per_s']/1e6))" | tee -a $P
    done
  done; done
  cp /tmp/libmpc_keep.so carnd-mpc-project_amd/lib/libmpc_amd.so
  ;;
li)   # instruction-cache counters of the builds in gpurun_in/ (a 6 % swing of the headline between two builds whose sweeps are identical)
  cp carnd-mpc-project_amd/lib/libmpc_amd.so /tmp/libmpc_keep.so
  cd /tmp && export TMPDIR=/tmp
  for f in $R/gpurun_in/libmpc_*.so; do
    cp $f $R/carnd-mpc-project_amd/lib/libmpc_amd.so; t=$(basename $f .so)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/r03li_$t -o pmc -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-legs > /dev/null 2> $OUT/r03li_$t.err; echo "$t exit=$?" | tee -a $P
  done
  cp /tmp/libmpc_keep.so $R/carnd-mpc-project_amd/lib/libmpc_amd.so
  cd $R
  for f in gpurun_in/libmpc_*.so; do t=$(basename $f .so); python tools/pmc_mean.py $OUT/r03li_$t | tee -a $P; done
  ;;
ip)   # fewer waves than instances / 64 (MPC_INSTANCES_PER_LANE): lanes take instances in turn from the start
  for rep in a b; do for ipl in 1 2 3; do for nfl in 2 3; do
    MPC_INSTANCES_PER_LANE=$ipl timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --steps 100 --inflight $nfl 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   headline per_lane=$ipl inflight=$nfl $rep  %.3f M' % (r['value']/1e6))" | tee -a $P
  done; done; done
  for ipl in 1 2; do for rm in "16 8" "8 4" "32 12"; do set -- $rm
    MPC_INSTANCES_PER_LANE=$ipl MPC_REFILL_MIN=$1 MPC_REFILL_WAIT=$2 timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --steps 100 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   headline per_lane=$ipl refill_min=$1 wait=$2  %.3f M' % (r['value']/1e6))" | tee -a $P
  done; done
  ;;
nx)   # where does the fp32 start begin to pay?  horizon sweep at a constant horizon time of 1.25 s, 32 768 instances
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --config config-stable.json --batch 32768 --steps 60 "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   %-28s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s' % ('$tag', r['value']/1e6, r['ms_per_step'], r['mean_iterations'], r['max_iterations'], {k: v for k, v in r['status_counts'].items() if v}))" | tee -a $P; }
  for N in 10 12 15 20 25 40; do
    dt=$(python -c "print(1.25 / $N)")
    for nfl in 2 4; do run N${N}_plain_i$nfl --N $N --dt $dt --inflight $nfl; done
    for nfl in 3 8; do run N${N}_f32start_i$nfl --N $N --dt $dt --inflight $nfl --f64-f32-start; done
  done
  ;;
mt)   # mixed-precision legs with deferred tails (an instance the fp64 phase cannot finish is solved again: a long chain)
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s' % ('$tag', r['value']/1e6, r['ms_per_step'], r['mean_iterations'], r['max_iterations'], {k: v for k, v in r['status_counts'].items() if v}))" | tee -a $P; }
  N25="--config config-stable.json --N 25 --dt 0.05 --batch 32768 --steps 80 --f64-f32-start"
  for c in 0 12 16 24 32; do for nfl in 4 8; do run n25_f32s_c${c}_i$nfl $N25 --inflight $nfl --tail-cut $c; done; done
  W="--weights-sweep --precision f32 --no-traj --batch 131072 --steps 60"
  for c in 0 12 16 24 32; do for nfl in 4 8; do run w32_mixed_c${c}_i$nfl $W --inflight $nfl --tail-cut $c; done; done
  ;;
mu)   # the N = 25 share as shipped (fp32 start) with deferred tails: run length
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s' % ('$tag', r['value']/1e6, r['ms_per_step'], r['mean_iterations'], r['max_iterations'], {k: v for k, v in r['status_counts'].items() if v}))" | tee -a $P; }
  N25="--config config-stable.json --N 25 --dt 0.05 --batch 32768 --f64-f32-start --inflight 8"
  for st in 80 400 1000; do for c in 12 16 20; do run n25_f32s_c${c}_s$st $N25 --steps $st --tail-cut $c --tail-ring 64; done; done
  run n25_f32s_c0_s400 $N25 --steps 400
  ;;
pb)   # mixed precision: the promoted iterates in a buffer of their own (the fp32 phase's lanes are free after the hand-over) against in their columns
  timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "f32 or mixed or lane_compaction or deferred_tails_are_bitwise or soak" > $OUT/r03pb_pytest.log 2>&1; rc=$?; echo "pytest exit=$rc" | tee -a $P; tail -5 $OUT/r03pb_pytest.log
  if [ $rc -ne 0 ]; then exit 1; fi
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s' % ('$tag', r['value']/1e6, r['ms_per_step'], r['mean_iterations'], r['max_iterations'], {k: v for k, v in r['status_counts'].items() if v}))" | tee -a $P; }
  for rep in a b; do
    run plain_$rep --steps 100
    for pb in 0 1; do for nfl in 2 3 4; do MPC_PROMOTE_BUFFER=$pb run head_f32s_pb${pb}_i${nfl}_$rep --steps 100 --f64-f32-start --inflight $nfl; done; done
    for pb in 0 1; do MPC_PROMOTE_BUFFER=$pb run n25_f32s_pb${pb}_$rep --config config-stable.json --N 25 --dt 0.05 --batch 32768 --f64-f32-start --inflight 8 --steps 400 --tail-cut 12 --tail-ring 64; done
    for pb in 0 1; do MPC_PROMOTE_BUFFER=$pb run w32_mixed_pb${pb}_$rep --weights-sweep --precision f32 --no-traj --batch 131072 --steps 100 --inflight 8 --tail-cut 12; done
  done
  ;;
rf)   # refill floor: no further takes once fewer lanes than this are running (plain solve; fp32 phase of a mixed solve with the promote buffer)
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d' % ('$tag', r['value']/1e6, r['ms_per_step'], r['mean_iterations'], r['max_iterations']))" | tee -a $P; }
  for fl in 0 8 16 32 48; do MPC_REFILL_FLOOR=$fl run plain_floor$fl --steps 100; done
  MPC_PROMOTE_BUFFER=0 run head_f32s_pb0 --steps 100 --f64-f32-start --inflight 3
  for fl in 0 16 32 48 64; do MPC_REFILL_FLOOR_F32=$fl run head_f32s_floor$fl --steps 100 --f64-f32-start --inflight 3; done
  MPC_PROMOTE_BUFFER=0 run n25_f32s_pb0 --config config-stable.json --N 25 --dt 0.05 --batch 32768 --f64-f32-start --inflight 8 --steps 400 --tail-cut 12 --tail-ring 64
  for fl in 0 16 32 48 64; do MPC_REFILL_FLOOR_F32=$fl run n25_f32s_floor$fl --config config-stable.json --N 25 --dt 0.05 --batch 32768 --f64-f32-start --inflight 8 --steps 400 --tail-cut 12 --tail-ring 64; done
  MPC_PROMOTE_BUFFER=0 run w32_mixed_pb0 --weights-sweep --precision f32 --no-traj --batch 131072 --steps 100 --inflight 8 --tail-cut 12
  for fl in 0 16 32 48 64; do MPC_REFILL_FLOOR_F32=$fl run w32_mixed_floor$fl --weights-sweep --precision f32 --no-traj --batch 131072 --steps 100 --inflight 8 --tail-cut 12; done
  ;;
rg)   # what decides whether the promote buffer pays: batch size, weights, precision at the ABI
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   %-30s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d' % ('$tag', r['value']/1e6, r['ms_per_step'], r['mean_iterations'], r['max_iterations']))" | tee -a $P; }
  for pb in 0 1; do
    export MPC_PROMOTE_BUFFER=$pb
    run f32io_plainw_B65536_pb$pb --precision f32 --no-traj --batch 65536 --steps 100 --inflight 4
    run f32io_plainw_B131072_pb$pb --precision f32 --no-traj --batch 131072 --steps 100 --inflight 4
    run f32io_sweep_B65536_pb$pb --weights-sweep --precision f32 --no-traj --batch 65536 --steps 100 --inflight 8 --tail-cut 12
    run f32io_sweep_B131072_i4_notail_pb$pb --weights-sweep --precision f32 --no-traj --batch 131072 --steps 100 --inflight 4
    run f64io_sweep_B65536_pb$pb --weights-sweep --batch 65536 --steps 100 --inflight 4 --f64-f32-start --tail-cut 12
    run f64io_head_B131072_pb$pb --batch 131072 --steps 60 --inflight 3 --f64-f32-start
  done
  unset MPC_PROMOTE_BUFFER
  ;;
sx)   # the whole gpu suite under the aggressive settings of the opt-in / tunable machinery
  MPC_LANE_COMPACT=1 MPC_LANE_COMPACT_COOLDOWN=0 timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/r03sx_compact.log 2>&1; echo "lane_compact 1 / cooldown 0: exit=$?" | tee -a $P; tail -4 $OUT/r03sx_compact.log
  MPC_PROMOTE_BUFFER=1 timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/r03sx_buffer.log 2>&1; echo "promote buffer forced: exit=$?" | tee -a $P; tail -4 $OUT/r03sx_buffer.log
  MPC_MIXED=1 MPC_PROMOTE_BUFFER=1 timeout -k 10 1000 python -m pytest tests -m gpu -q -k "matches_oracle or scipy or full_size_properties or soak or test_cpp or horizon_extremes or rollout or plot_anchors or run_path or leave_the_central_path" > $OUT/r03sx_mixed.log 2>&1; echo "fp32 start forced on every fp64 handle + buffer: exit=$?" | tee -a $P; tail -6 $OUT/r03sx_mixed.log
  ;;
uf)   # the unfiltered population with the fp32 start (+ refill) and deferred tails
  run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --population survey --tail-ring 64 --steps 2000 "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s' % ('$tag', r['value']/1e6, r['ms_per_step'], r['mean_iterations'], r['max_iterations'], {k: v for k, v in r['status_counts'].items() if v}))" | tee -a $P; }
  run plain_c20 --tail-cut 20
  for c in 8 12 16; do run f32s_c${c}_i3 --f64-f32-start --inflight 3 --tail-cut $c; done
  for c in 8 12 16; do run f32s_refill_c${c}_i3 --f64-f32-start --f32-phase-refill --inflight 3 --tail-cut $c; done
  run f32s_refill_c12_i4 --f64-f32-start --f32-phase-refill --inflight 4 --tail-cut 12
  ;;
n3)   # configs[3] share drawn with SURVEY's rejection only: single phase against the fp32 start, deferred tails
  run() { tag=$1; shift; timeout -k 10 400 python bench.py --no-legs --no-cpu-baseline --no-host-leg --population survey --tail-ring 64 --config config-stable.json --N 25 --dt 0.05 --batch 32768 "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s' % ('$tag', r['value']/1e6, r['ms_per_step'], r['mean_iterations'], r['max_iterations'], {k: v for k, v in r['status_counts'].items() if v}))" | tee -a $P; }
  run plain_c0_i4 --steps 40 --inflight 4
  for c in 16 24 32; do run plain_c${c}_i4 --steps 400 --inflight 4 --tail-cut $c; done
  run f32s_c0_i8 --steps 40 --inflight 8 --f64-f32-start
  for c in 12 16 24; do run f32s_c${c}_i8 --steps 400 --inflight 8 --f64-f32-start --tail-cut $c; done
  run f32s_refill_c12_i8 --steps 400 --inflight 8 --f64-f32-start --f32-phase-refill --tail-cut 12
  ;;
n4)   # configs[4] share drawn with SURVEY's rejection only
  run() { tag=$1; shift; timeout -k 10 400 python bench.py --no-legs --no-cpu-baseline --no-host-leg --population survey --tail-ring 64 --weights-sweep --precision f32 --no-traj --batch 131072 "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   %-26s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s' % ('$tag', r['value']/1e6, r['ms_per_step'], r['mean_iterations'], r['max_iterations'], {k: v for k, v in r['status_counts'].items() if v}))" | tee -a $P; }
  run mixed_c12_refill_i8 --steps 150 --inflight 8 --tail-cut 12 --f32-phase-refill
  run mixed_c12_i8 --steps 150 --inflight 8 --tail-cut 12
  run mixed_c0_i4 --steps 40 --inflight 4
  run pure_c24_i4 --steps 150 --inflight 4 --tail-cut 24 --f32-pure
  run pure_c0_i4 --steps 40 --inflight 4 --f32-pure
  timeout -k 10 300 python bench.py --leg configs_3_share_unfiltered 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   leg configs_3_share_unfiltered  %.3f M  %s' % (r['solves_per_s']/1e6, r['status_counts']))" | tee -a $P
  ;;
pc)   # mixed precision with the clean-hand-over rule: the fp32 phase's iteration allowance (MPC_PROMOTE_CAP)
  timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "f32 or mixed or deferred_tails_are_bitwise or central_path" > $OUT/r03pc_pytest.log 2>&1; rc=$?; echo "pytest exit=$rc" | tee -a $P; tail -5 $OUT/r03pc_pytest.log
  run() { tag=$1; shift; timeout -k 10 400 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   %-28s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s' % ('$tag', r['value']/1e6, r['ms_per_step'], r['mean_iterations'], r['max_iterations'], {k: v for k, v in r['status_counts'].items() if v}))" | tee -a $P; }
  run head_plain --steps 100
  for cap in 40 24 16 12; do
    export MPC_PROMOTE_CAP=$cap
    run head_f32s_cap$cap --steps 100 --f64-f32-start --inflight 3
    run n25_f32s_cap$cap --config config-stable.json --N 25 --dt 0.05 --batch 32768 --f64-f32-start --inflight 8 --steps 400 --tail-cut 12 --tail-ring 64
    run n25_f32s_notail_cap$cap --config config-stable.json --N 25 --dt 0.05 --batch 32768 --f64-f32-start --inflight 8 --steps 80
    run w32_mixed_cap$cap --weights-sweep --precision f32 --no-traj --batch 131072 --steps 100 --inflight 8 --tail-cut 12 --f32-phase-refill
    run survey_f32s_c20_cap$cap --population survey --tail-ring 64 --steps 1000 --f64-f32-start --inflight 3 --tail-cut 20
    run n25_survey_f32s_cap$cap --population survey --tail-ring 64 --config config-stable.json --N 25 --dt 0.05 --batch 32768 --steps 400 --inflight 8 --f64-f32-start --tail-cut 24
  done
  unset MPC_PROMOTE_CAP
  ;;
ph)   # the headline with the fp32 start under the clean-hand-over rule: its restarted instances as deferred tails
  run() { tag=$1; shift; timeout -k 10 400 python bench.py --no-legs --no-cpu-baseline --no-host-leg "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   %-28s %8.3f M solves/s  %.3f ms/batch  iters %.2f max %d  status %s' % ('$tag', r['value']/1e6, r['ms_per_step'], r['mean_iterations'], r['max_iterations'], {k: v for k, v in r['status_counts'].items() if v}))" | tee -a $P; }
  run head_plain --steps 200
  for c in 0 6 8 10 12; do for nfl in 3 4; do run head_f32s_c${c}_i$nfl --steps 1000 --f64-f32-start --inflight $nfl --tail-cut $c --tail-ring 64; done; done
  ;;
esac
echo done | tee -a $P
