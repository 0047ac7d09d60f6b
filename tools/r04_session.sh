#!/bin/bash
# The GPU sessions of round 4 (one parameterised script; run from the repo root through gpurun):
#   bash tools/r04_session.sh a     survey population: tail waves x tail cut; N = 25 PMC passes
set -o pipefail
S=${1:-a}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
P=$OUT/r04${S}_progress.log
echo "== start $S" | tee $P
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --no-host-leg --full-json $OUT/r04${S}_$tag.json "$@" > $OUT/r04${S}_$tag.line 2> $OUT/r04${S}_$tag.err; echo "$tag exit=$?" | tee -a $P; python - <<PY | tee -a $P
import json
try:
    r = json.load(open("$OUT/r04${S}_$tag.json"))
    t = r["config"]["deferred_tails"]
    print("   %-26s %8.3f M solves/s (strict %.2f M)  %.3f ms/batch  iters %.2f max %d  status %s  kernel_ms %.3f  timing %s  tails %s" % ("$tag", r["value"] / 1e6, r["strict_value"] / 1e6, r["ms_per_step"], r["mean_iterations"], r["max_iterations"], {k: v for k, v in r["status_counts"].items() if v}, r["roofline"]["kernel_ms_avg"], [r["timing"][k] for k in ("first_timed_batch", "batches_in_the_window", "batches_outstanding_at_the_end")], t if isinstance(t, str) else {k: t[k] for k in ("tail_slices", "batches_deferred", "tail_cut_in_use", "batches_not_deferred_survivors_full")}))
except Exception as e:
    print("   $tag: no result", e)
PY
}
pmc() { tag=$1; ctrs=$2; shift 2; ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/r04${S}_pmc_$tag -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-host-leg --no-legs "$@" > /dev/null 2> $OUT/r04${S}_pmc_$tag.err ); echo "pmc $tag exit=$?" | tee -a $P; }
case $S in
a)
  export GPU_MAX_HW_QUEUES=8
  run head_c0 --steps 200
  for w in 8 16 32 64 256; do MPC_TAIL_WAVES=$w run survey_c20_w$w --steps 600 --population survey --tail-cut 20 --tail-ring 64; done
  for c in 24 28 32 40; do MPC_TAIL_WAVES=32 run survey_c${c}_w32 --steps 600 --population survey --tail-cut $c --tail-ring 64; done
  for c in 24 32; do MPC_TAIL_WAVES=16 run survey_c${c}_w16 --steps 600 --population survey --tail-cut $c --tail-ring 64; done
  MPC_TAIL_WAVES=32 MPC_TAIL_STREAMS=1 run survey_c24_w32_st1 --steps 600 --population survey --tail-cut 24 --tail-ring 64
  MPC_TAIL_WAVES=32 MPC_TAIL_STREAMS=3 run survey_c24_w32_st3 --steps 600 --population survey --tail-cut 24 --tail-ring 64
  MPC_TAIL_WAVES=32 MPC_TAIL_PRIORITY=normal run survey_c24_w32_np --steps 600 --population survey --tail-cut 24 --tail-ring 64
  # N = 25 (configs[3] share): r04 PMC passes of the single-phase fp64 solve
  N25="--N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4 --tail-cut 24 --tail-ring 64"
  run n25 --steps 60 $N25
  MPC_TAIL_WAVES=32 run n25_w32 --steps 60 $N25
  MPC_LANE_COMPACT=1 run n25_lc1 --steps 60 $N25
  pmc n25_fetch FETCH_SIZE --steps 6 --warmup 4 $N25
  pmc n25_write WRITE_SIZE --steps 6 --warmup 4 $N25
  pmc n25_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --steps 6 --warmup 4 $N25
  ;;
b)   # round 4's tail slices: parity (bitwise) tests first, then the survey population
  export GPU_MAX_HW_QUEUES=8
  timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "deferred or lane_compaction or phase_refill or native or test_cpp or edge_cases or batch_matches" > $OUT/r04b_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -8 $OUT/r04b_pytest.log
  run filtered_c0 --steps 100 --population filtered --tail-cut 0
  run survey_auto --steps 200
  run survey_auto_k20 --steps 20
  for c in 12 16 20 24 32; do run survey_c$c --steps 200 --tail-cut $c; done
  for sp in 12 36; do MPC_SLICE_PASSES=$sp run survey_c20_sp$sp --steps 200 --tail-cut 20; done
  MPC_TAIL_WAVES=32 run survey_c20_w32 --steps 200 --tail-cut 20
  MPC_TAIL_PRIORITY=normal run survey_c20_np --steps 200 --tail-cut 20
  run survey_c20_i3 --steps 200 --tail-cut 20 --inflight 3
  ;;
c)   # tail slices, tuned: the bench line with and without legs
  export GPU_MAX_HW_QUEUES=8
  run survey_auto_k20 --steps 20
  run survey_auto_k200 --steps 200
  run survey_auto_k1000 --steps 1000
  run filtered_c0 --steps 100 --population filtered --tail-cut 0
  ( time timeout -k 10 900 python bench.py --full-json $OUT/r04c_bench_full.json > $OUT/r04c_bench.line 2> $OUT/r04c_bench.err ) 2>&1 | tail -3 | tee -a $P; cat $OUT/r04c_bench.line | tee -a $P
  ;;
d)   # the whole GPU suite, then slice length and the K = 20 line
  export GPU_MAX_HW_QUEUES=8
  timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/r04d_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -6 $OUT/r04d_pytest.log
  for sp in 12 16 32; do MPC_SLICE_PASSES=$sp run survey_sp$sp --steps 200; done
  for k in 1 2 3; do run survey_k20_$k --steps 20 --warmup 5; done
  ;;
e)   # fixed tests, the default line, the one-rank collective rehearsal, N = 25
  export GPU_MAX_HW_QUEUES=8
  timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "tail_wait_resolves or run_batch_host or lane_compaction or fp32_start or drop_in" > $OUT/r04e_pytest2.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -4 $OUT/r04e_pytest2.log
  ( time timeout -k 10 900 python bench.py --steps 20 --warmup 5 --full-json $OUT/r04e_bench_full.json > $OUT/r04e_bench.line 2> $OUT/r04e_bench.err ) 2>&1 | tail -3 | tee -a $P; wc -c $OUT/r04e_bench.line | tee -a $P
  run coll_root --steps 200 --force-collective
  run coll_root_g1 --steps 200 --force-collective --gather-group 1
  run coll_all --steps 200 --force-collective --gather all
  run coll_none --steps 200
  N25="--N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4"
  run n25_auto --steps 60 $N25
  run n25_i6 --steps 60 --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 6
  MPC_LANE_COMPACT=2 run n25_lc2 --steps 60 $N25
  MPC_TAIL_FEW=8 run n25_few8 --steps 60 $N25
  MPC_TAIL_AUTO_CUT=20 run n25_c20 --steps 60 $N25
  MPC_TAIL_AUTO_CUT=28 run n25_c28 --steps 60 $N25
  run n25_f32start --steps 60 $N25 --f64-f32-start
  ;;
f)   # direct remote-write gather; the default line; rocprof of the headline
  export GPU_MAX_HW_QUEUES=8
  timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "direct_gather or tail_wait_resolves" > $OUT/r04f_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -4 $OUT/r04f_pytest.log
  run coll_direct --steps 200 --force-collective
  run coll_none --steps 200
  run coll_root --steps 200 --force-collective --gather root
  MPC_TAIL_AUTO_CUT=16 run n25_c16_adapts --steps 60 --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4
  ( time timeout -k 10 900 python bench.py --steps 20 --warmup 5 --full-json $OUT/r04f_bench_full.json > $OUT/r04f_bench.line 2> $OUT/r04f_bench.err ) 2>&1 | tail -3 | tee -a $P; wc -c $OUT/r04f_bench.line | tee -a $P
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04f_prof -o trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-legs --full-json $OUT/r04f_prof_bench.json > /dev/null 2> $OUT/r04f_prof.err; echo "rocprof exit=$?" | tee -a $P
  for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/r04f_pmc_$c -o pmc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-legs --full-json $OUT/r04f_pmc_$c.json > /dev/null 2> $OUT/r04f_pmc_$c.err; echo "pmc $c exit=$?" | tee -a $P; done
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/r04f_pmc_sq -o pmc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-legs --full-json $OUT/r04f_pmc_sq.json > /dev/null 2> $OUT/r04f_pmc_sq.err; echo "pmc sq exit=$?" | tee -a $P
  cd $R
  ;;
g)   # the legs again with the adaptive cut
  export GPU_MAX_HW_QUEUES=8
  for l in ${LEGS_G:-configs_4_share configs_4_share_filtered configs_3_share_f32_start configs_3_share_filtered}; do
    timeout -k 10 300 python bench.py --leg $l --tail-ring 128 > $OUT/r04g_$l.json 2> $OUT/r04g_$l.err; echo "$l exit=$?" | tee -a $P
    python - <<PY | tee -a $P
import json
try:
    l = json.load(open("$OUT/r04g_$l.json"))
    print("   %-28s %7.2f M (strict %6.2f) iters %.2f max %d status %s tails %s" % ("$l", l["solves_per_s"]/1e6, l["strict_solves_per_s"]/1e6, l["mean_iterations"], l["max_iterations"], {a:b for a,b in l["status_counts"].items() if b}, {k: l["tails"][k] for k in ("tail_cut_in_use", "queue_overflows", "batches_not_deferred_survivors_full", "tail_slices")}))
except Exception as e:
    print("   $l: no result", e)
PY
  done
  ;;
h)   # the fp32 start on the filtered and on SURVEY's populations, with the tail slices
  export GPU_MAX_HW_QUEUES=8
  N25="--N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4"
  for pop in filtered survey; do
    run n10_${pop}_plain --steps 200 --population $pop
    run n10_${pop}_f32start --steps 200 --population $pop --f64-f32-start
    run n25_${pop}_plain --steps 100 --population $pop $N25
    run n25_${pop}_f32start --steps 100 --population $pop $N25 --f64-f32-start
  done
  ;;
i)   # N = 25 on SURVEY's population, long windows: plain against the fp32 start
  export GPU_MAX_HW_QUEUES=8
  N25="--N 25 --dt 0.05 --config config-stable.json --batch 32768"
  run n25_survey_plain_i4 --steps 500 $N25 --inflight 4
  run n25_survey_f32start_i4 --steps 500 $N25 --inflight 4 --f64-f32-start
  run n25_survey_f32start_i8 --steps 500 $N25 --inflight 8 --f64-f32-start
  run n25_filtered_plain_i4 --steps 500 $N25 --inflight 4 --population filtered
  run n25_filtered_f32start_i4 --steps 500 $N25 --inflight 4 --population filtered --f64-f32-start
  run n25_filtered_f32start_i8 --steps 500 $N25 --inflight 8 --population filtered --f64-f32-start
  ;;
j)   # N = 25 on SURVEY's population, long windows: what keeps the slices from keeping up
  export GPU_MAX_HW_QUEUES=8
  N25="--N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4"
  MPC_TAIL_PRIORITY=high run n25_prio_high --steps 400 $N25
  MPC_SLICE_PASSES=32 run n25_sp32 --steps 400 $N25
  MPC_TAIL_PRIORITY=high MPC_SLICE_PASSES=32 run n25_prio_high_sp32 --steps 400 $N25
  MPC_TAIL_AUTO_CUT=32 run n25_cut32 --steps 400 $N25
  MPC_TAIL_PRIORITY=high MPC_TAIL_AUTO_CUT=32 run n25_prio_high_cut32 --steps 400 $N25
  MPC_TAIL_PRIORITY=high MPC_TAIL_WAVES=512 run n25_prio_high_w512 --steps 400 $N25
  MPC_TAIL_PRIORITY=high run n25_prio_high_i2 --steps 400 --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 2
  MPC_TAIL_PRIORITY=high run n10_prio_high --steps 400
  run n10_plain --steps 400
  ;;
k)   # N = 25 on SURVEY's population, tail stream high priority: waves per slice, passes per slice, the cut
  export GPU_MAX_HW_QUEUES=8 MPC_TAIL_PRIORITY=high
  N25="--N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4"
  MPC_TAIL_AUTO_CUT=32 MPC_SLICE_FRESH_DIV=1 run n25_c32_fd1 --steps 400 $N25
  MPC_TAIL_AUTO_CUT=32 MPC_SLICE_FRESH_DIV=2 run n25_c32_fd2 --steps 400 $N25
  MPC_TAIL_AUTO_CUT=32 MPC_SLICE_PASSES=8 run n25_c32_sp8 --steps 400 $N25
  MPC_TAIL_AUTO_CUT=32 MPC_SLICE_PASSES=24 run n25_c32_sp24 --steps 400 $N25
  MPC_TAIL_AUTO_CUT=40 run n25_c40 --steps 400 $N25
  MPC_TAIL_AUTO_CUT=48 run n25_c48 --steps 400 $N25
  MPC_TAIL_AUTO_CUT=32 MPC_TAIL_FEW=0 run n25_c32_few0 --steps 400 $N25
  MPC_TAIL_AUTO_CUT=32 MPC_TAIL_WAVES=512 run n25_c32_w512 --steps 400 $N25
  MPC_TAIL_AUTO_CUT=32 run n25_c32_f32start --steps 400 $N25 --f64-f32-start
  ;;
l)   # what the slices do at N = 25 and N = 10 (MPC_TAIL_TRACE)
  export GPU_MAX_HW_QUEUES=8
  N25="--N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4"
  rm -f $OUT/r04l_trace_*.txt
  MPC_TAIL_TRACE=$OUT/r04l_trace_n25.txt run n25_trace --steps 400 $N25
  python tools/slice_trace.py $OUT/r04l_trace_n25.txt 100 | tee -a $P
  MPC_TAIL_PRIORITY=high MPC_TAIL_TRACE=$OUT/r04l_trace_n25_high.txt run n25_trace_high --steps 400 $N25
  python tools/slice_trace.py $OUT/r04l_trace_n25_high.txt 100 | tee -a $P
  MPC_TAIL_TRACE=$OUT/r04l_trace_n10.txt run n10_trace --steps 400
  python tools/slice_trace.py $OUT/r04l_trace_n10.txt 100 | tee -a $P
  ;;
m)   # N = 25 on SURVEY's population: buffer sets (batches outstanding), tail stream priority, the cut
  export GPU_MAX_HW_QUEUES=8
  N25="--N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4"
  MPC_TAIL_PRIORITY=high MPC_TAIL_AUTO_CUT=32 run n25_high_c32_o512 --steps 600 $N25 --outstanding 512
  MPC_TAIL_PRIORITY=high MPC_TAIL_AUTO_CUT=24 run n25_high_c24_o512 --steps 600 $N25 --outstanding 512
  MPC_TAIL_PRIORITY=high MPC_TAIL_AUTO_CUT=32 run n25_high_c32_o256 --steps 600 $N25
  MPC_TAIL_PRIORITY=high MPC_TAIL_AUTO_CUT=24 run n25_high_c24_o256 --steps 600 $N25
  MPC_TAIL_PRIORITY=normal MPC_TAIL_AUTO_CUT=32 run n25_norm_c32_o512 --steps 600 $N25 --outstanding 512
  MPC_TAIL_PRIORITY=high MPC_TAIL_AUTO_CUT=32 run n25_high_c32_o512_f32start --steps 600 $N25 --outstanding 512 --f64-f32-start
  MPC_TAIL_PRIORITY=high run n10_high_o512 --steps 600 --outstanding 512
  ;;
n)   # the whole GPU suite with MPC_F32_START_AUTO as the default and the tail stream on high priority; the N = 25 legs
  export GPU_MAX_HW_QUEUES=8
  timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/r04n_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -6 $OUT/r04n_pytest.log
  LEGS_G="configs_3_share configs_3_share_filtered configs_3_share_fp64_only" bash tools/r04_session.sh g
  ;;
p)   # the round's evidence on the final build: full-scale soak, kernel trace and PMC passes of the driver's command
  export GPU_MAX_HW_QUEUES=8
  rm -f $OUT/soak*.json
  ( time MPC_SOAK=1 MPC_SOAK_WORKERS=60 timeout -k 10 900 python -m pytest tests/test_soak.py -m gpu -q -x > $OUT/r04p_soak.log 2>&1 ) 2>&1 | tail -3 | tee -a $P; tail -3 $OUT/r04p_soak.log | tee -a $P
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04p_prof -o trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-legs --full-json $OUT/r04p_prof_bench.json > /dev/null 2> $OUT/r04p_prof.err; echo "rocprof exit=$?" | tee -a $P
  for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/r04p_pmc_$c -o pmc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-legs --full-json $OUT/r04p_pmc_$c.json > /dev/null 2> $OUT/r04p_pmc_$c.err; echo "pmc $c exit=$?" | tee -a $P; done
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/r04p_pmc_sq -o pmc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-legs --full-json $OUT/r04p_pmc_sq.json > /dev/null 2> $OUT/r04p_pmc_sq.err; echo "pmc sq exit=$?" | tee -a $P
  cd $R
  ;;
q)   # the initial state's own rows carried (lam_0, z_0): the whole GPU suite, then the rates
  export GPU_MAX_HW_QUEUES=8
  timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/r04q_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -5 $OUT/r04q_pytest.log
  run survey_k400 --steps 400
  run filtered_c0 --steps 200 --population filtered --tail-cut 0
  run survey_k400_b --steps 400
  run n25_survey --steps 400 --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 4
  ;;
s)   # GPU suite after the initial_state_rows switch, then rates
  export GPU_MAX_HW_QUEUES=8
  timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/r04s_pytest.log 2>&1; echo "pytest exit=$?" | tee -a $P; tail -5 $OUT/r04s_pytest.log
  run filtered_c0 --steps 200 --population filtered --tail-cut 0
  run survey_k400 --steps 400
  run filtered_c0_rows --steps 200 --population filtered --tail-cut 0 --initial-state-rows
  ;;
v)   # the fp32 phase of the long-horizon solve at two waves per SIMD
  export GPU_MAX_HW_QUEUES=8
  N25="--N 25 --dt 0.05 --config config-stable.json --batch 32768"
  run n25_occ1_i4 --steps 400 $N25 --inflight 4
  MPC_F32_OCC=2 run n25_occ2_i4 --steps 400 $N25 --inflight 4
  run n25_occ1_i8 --steps 400 $N25 --inflight 8
  MPC_F32_OCC=2 run n25_occ2_i8 --steps 400 $N25 --inflight 8
  MPC_F32_OCC=2 run n25_occ2_i8_filtered --steps 400 $N25 --inflight 8 --population filtered
  run n25_occ1_i8_filtered --steps 400 $N25 --inflight 8 --population filtered
  ;;
w)   # the legs with the library's in-flight advice; the fp64 weight sweep with 2 and 4 in flight
  export GPU_MAX_HW_QUEUES=8
  for n in 2 4; do
    timeout -k 10 300 python bench.py --leg weights_sweep_f64 --leg-inflight $n > $OUT/r04w_ws_$n.json 2> $OUT/r04w_ws_$n.err; echo "weights_sweep_f64 inflight $n exit=$?" | tee -a $P
    python -c "import json; l = json.load(open('$OUT/r04w_ws_$n.json')); print('   %.2f M (strict %.2f) in flight %d advised %d' % (l['solves_per_s'] / 1e6, l['strict_solves_per_s'] / 1e6, l['batches_in_flight'], l['batches_in_flight_advised']))" | tee -a $P
  done
  run head_default --steps 200
  ;;
z)   # where the fp32 start pays at N = 10: launches that do not fill the device
  export GPU_MAX_HW_QUEUES=8
  for b in 4096 16384 32768; do
    run n10_b${b}_plain --steps 400 --batch $b --tail-cut 0 --population filtered
    run n10_b${b}_f32start --steps 400 --batch $b --tail-cut 0 --population filtered --f64-f32-start
  done
  ;;
y)   # batches in flight for launches of 8 192 .. 32 768 instances (N = 10, single-phase fp64)
  export GPU_MAX_HW_QUEUES=8
  for b in 8192 16384 32768; do
    for n in 2 4 8; do run n10_b${b}_i$n --steps 400 --batch $b --tail-cut 0 --population filtered --inflight $n; done
  done
  run n10_b65536_i4 --steps 200 --tail-cut 0 --population filtered --inflight 4
  ;;
x)   # small launches with 8 / 16 / 32 in flight
  export GPU_MAX_HW_QUEUES=8
  for b in 4096 8192; do
    for n in 8 16 32; do run n10_b${b}_i$n --steps 600 --batch $b --tail-cut 0 --population filtered --inflight $n; done
  done
  run n10_b1024_i16 --steps 600 --batch 1024 --tail-cut 0 --population filtered --inflight 16
  run n10_b1024_i32 --steps 600 --batch 1024 --tail-cut 0 --population filtered --inflight 32
  ;;
u)   # long windows on the final build: the pipeline over several seconds
  export GPU_MAX_HW_QUEUES=8
  run survey_k3000 --steps 3000
  run n25_k1500 --steps 1500 --N 25 --dt 0.05 --config config-stable.json --batch 32768
  timeout -k 10 300 python bench.py --leg configs_4_share --leg-steps 1000 > $OUT/r04u_c4.json 2> $OUT/r04u_c4.err; echo "configs_4_share exit=$?" | tee -a $P
  python -c "import json; l = json.load(open('$OUT/r04u_c4.json')); print('   configs_4_share %.2f M (strict %.2f) %s in flight %d' % (l['solves_per_s'] / 1e6, l['strict_solves_per_s'] / 1e6, l['status_counts'], l['batches_in_flight']))" | tee -a $P
  ;;
o)   # small launches in flight: the one-instance-per-wavefront kernel (default up to 1 024) against the lane kernel
  export GPU_MAX_HW_QUEUES=8
  for b in 256 1024; do
    run n10_b${b}_wave --steps 600 --batch $b --tail-cut 0 --population filtered
    MPC_WAVE_MAX_BATCH=0 run n10_b${b}_lane --steps 600 --batch $b --tail-cut 0 --population filtered
  done
  MPC_WAVE_MAX_BATCH=4096 run n10_b4096_wave --steps 400 --batch 4096 --tail-cut 0 --population filtered
  run n10_b4096_lane --steps 400 --batch 4096 --tail-cut 0 --population filtered
  ;;
r)   # rates only
  export GPU_MAX_HW_QUEUES=8
  run filtered_c0 --steps 200 --population filtered --tail-cut 0
  run survey_k400 --steps 400
  run filtered_c0_b --steps 200 --population filtered --tail-cut 0
  run filtered_c0_rows --steps 200 --population filtered --tail-cut 0 --initial-state-rows
  run survey_k400_rows --steps 400 --initial-state-rows
  ;;
esac
echo done | tee -a $P
