#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel trace + PMC passes.
# Usage (from the repo root, via gpurun):  bash tools/gpu_round.sh [tag]
set -o pipefail
TAG=${1:-r01}
BENCH_ARGS=${BENCH_ARGS:-}
SKIP_TESTS=${SKIP_TESTS:-0}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== start" | tee $OUT/${TAG}_progress.log
if [ "$SKIP_TESTS" != "1" ]; then
echo "== pytest -m gpu" | tee -a $OUT/${TAG}_progress.log
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit=$?" | tee -a $OUT/${TAG}_progress.log
tail -3 $OUT/${TAG}_pytest_gpu.log
fi
echo "== bench" | tee -a $OUT/${TAG}_progress.log
timeout -k 10 600 python bench.py $BENCH_ARGS > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit=$?" | tee -a $OUT/${TAG}_progress.log
cat $OUT/${TAG}_bench.json
echo "== rocprofv3 kernel trace" | tee -a $OUT/${TAG}_progress.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg --no-legs $BENCH_ARGS > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err; echo "rocprof exit=$?" | tee -a $OUT/${TAG}_progress.log
echo "== rocprofv3 pmc (separate passes)" | tee -a $OUT/${TAG}_progress.log
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-legs $BENCH_ARGS > /dev/null 2> $OUT/${TAG}_pmc_fetch.err; echo "pmc fetch exit=$?" | tee -a $OUT/${TAG}_progress.log
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-legs $BENCH_ARGS > /dev/null 2> $OUT/${TAG}_pmc_write.err; echo "pmc write exit=$?" | tee -a $OUT/${TAG}_progress.log
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_pmc_sq -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-legs $BENCH_ARGS > /dev/null 2> $OUT/${TAG}_pmc_sq.err; echo "pmc sq exit=$?" | tee -a $OUT/${TAG}_progress.log
echo "== FETCH/WRITE calibration at 8 B per lane" | tee -a $OUT/${TAG}_progress.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_fetch $R/tools/calib_fetch.hip > $OUT/${TAG}_calib_build.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_calib_fetch -o pmc -- /tmp/calib_fetch > $OUT/${TAG}_calib.log 2>&1; echo "calib fetch exit=$?" | tee -a $OUT/${TAG}_progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_calib_write -o pmc -- /tmp/calib_fetch >> $OUT/${TAG}_calib.log 2>&1; echo "calib write exit=$?" | tee -a $OUT/${TAG}_progress.log
find $OUT -name "*.csv" | grep ${TAG} | head -40
echo done | tee -a $OUT/${TAG}_progress.log
