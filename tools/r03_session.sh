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
esac
echo done | tee -a $P
