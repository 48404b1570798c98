#!/bin/bash
# One parametrised runner for every gpurun call of the round (replaces the per-run scripts of round 1):
#     gpurun --timeout 1500 -- 'bash scripts/gpu_run.sh <plan> [args...]'
# Every plan writes its logs under gpurun_out/ (merged back by gpurun) and a one-line verdict per step into
# gpurun_out/summary.log.  Plans:
#   tests [pytest args]      the -m gpu suite (default: all of tests/)
#   smoke                    __graft_entry__.smoke()
#   bench [bench.py args]    python bench.py ... -> gpurun_out/bench_<tag>.json  (tag = $BENCH_TAG, default "default")
#   prof  [bench.py args]    rocprofv3 --kernel-trace --stats of bench.py ... -> gpurun_out/prof_<tag>/
#   pmc <counters> <kernel> <windows/launch | auto> [bench.py args]   rocprofv3 --pmc (own pass) -> gpurun_out/pmc_<tag>_summary.json
#   pmcmfma [bench.py args]  MfmaUtil per kernel class (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE) -> gpurun_out/pmc_<tag>_summary.json
#   py <file> [args]         python <file> ... -> gpurun_out/<basename>.log
#   seq "<plan a...>" "<plan b...>"   several plans in one call
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG="${BENCH_TAG:-default}"
note() { echo "$*" >> gpurun_out/summary.log; }
plan="$1"; shift
case "$plan" in
  tests)
    args=("$@"); [ ${#args[@]} -eq 0 ] && args=(tests)
    timeout 2400 python -m pytest "${args[@]}" -m gpu -q --timeout 900 --maxfail=25 --durations=30 > gpurun_out/pytest_gpu.log 2>&1
    note "pytest ${args[*]} rc=$?"; tail -8 gpurun_out/pytest_gpu.log ;;
  smoke)
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
    note "smoke rc=$?"; tail -3 gpurun_out/smoke.log ;;
  bench)
    timeout ${BENCH_TIMEOUT:-600} python bench.py "$@" > "gpurun_out/bench_${TAG}.json" 2> "gpurun_out/bench_${TAG}.err"
    note "bench[$TAG] $* rc=$?"; cut -c1-1500 "gpurun_out/bench_${TAG}.json"; tail -3 "gpurun_out/bench_${TAG}.err" ;;
  prof)
    # rocprofv3 --kernel-trace --stats of bench.py; the raw trace (hundreds of MB for a 120-min step) is summarised on the
    # box into gpurun_out/prof_<tag>_kernel_stats.csv and removed (gpurun merges at most 64 MiB back)
    rm -rf "gpurun_out/prof_${TAG}"
    (cd /tmp && timeout ${BENCH_TIMEOUT:-900} rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_${TAG}" -o trace --output-format csv -- \
        python "$OLDPWD/bench.py" "$@" > "$OLDPWD/gpurun_out/prof_${TAG}.json" 2> "$OLDPWD/gpurun_out/prof_${TAG}.err")
    note "prof[$TAG] $* rc=$?"
    python scripts/rocprof_summarise.py stats "gpurun_out/prof_${TAG}" "gpurun_out/prof_${TAG}_kernel_stats.csv"; note "prof summary rc=$?"
    rm -rf "gpurun_out/prof_${TAG}"; cut -c1-400 "gpurun_out/prof_${TAG}.json" ;;
  pmc)
    # rocprofv3 --pmc <counters> in its OWN pass (no tracing domains): pmc <counters> <kernel substring> <windows/launch> [bench args]
    counters="$1"; kern="$2"; wpl="$3"; shift 3
    rm -rf "gpurun_out/pmc_${TAG}"
    (cd /tmp && timeout ${BENCH_TIMEOUT:-900} rocprofv3 --pmc $counters -d "$OLDPWD/gpurun_out/pmc_${TAG}" -o pmc --output-format csv -- \
        python "$OLDPWD/bench.py" "$@" > "$OLDPWD/gpurun_out/pmc_${TAG}.json" 2> "$OLDPWD/gpurun_out/pmc_${TAG}.err")
    note "pmc[$TAG] $counters $* rc=$?"
    [ "$wpl" = "auto" ] && wpl="auto:gpurun_out/pmc_${TAG}.json"
    python scripts/rocprof_summarise.py pmc "gpurun_out/pmc_${TAG}" "gpurun_out/pmc_${TAG}_summary.json" "$kern" "$wpl" python bench.py "$@"; note "pmc summary rc=$?"
    rm -rf "gpurun_out/pmc_${TAG}" ;;
  pmcmfma)
    # MfmaUtil per kernel class: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128), own pass with --kernel-trace only
    rm -rf "gpurun_out/pmc_${TAG}"
    (cd /tmp && timeout ${BENCH_TIMEOUT:-900} rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$OLDPWD/gpurun_out/pmc_${TAG}" -o pmc --output-format csv -- \
        python "$OLDPWD/bench.py" "$@" > "$OLDPWD/gpurun_out/pmc_${TAG}.json" 2> "$OLDPWD/gpurun_out/pmc_${TAG}.err")
    note "pmcmfma[$TAG] $* rc=$?"
    python scripts/rocprof_summarise.py mfma "gpurun_out/pmc_${TAG}" "gpurun_out/pmc_${TAG}_summary.json" python bench.py "$@"; note "pmcmfma summary rc=$?"
    rm -rf "gpurun_out/pmc_${TAG}" ;;
  pmcfetch)
    # FETCH_SIZE of every dispatch with the kernel trace beside it (own pass): per-kernel HBM reads and bytes per decode iteration (cfg5)
    rm -rf "gpurun_out/pmc_${TAG}"
    (cd /tmp && timeout ${BENCH_TIMEOUT:-900} rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OLDPWD/gpurun_out/pmc_${TAG}" -o pmc --output-format csv -- \
        python "$OLDPWD/bench.py" "$@" > "$OLDPWD/gpurun_out/pmc_${TAG}.json" 2> "$OLDPWD/gpurun_out/pmc_${TAG}.err")
    note "pmcfetch[$TAG] $* rc=$?"
    python scripts/rocprof_summarise.py fetch "gpurun_out/pmc_${TAG}" "gpurun_out/pmc_${TAG}_summary.json" "gpurun_out/pmc_${TAG}.json" python bench.py "$@"; note "pmcfetch summary rc=$?"
    rm -rf "gpurun_out/pmc_${TAG}" ;;
  py)
    f="$1"; shift
    timeout 1800 python "$f" "$@" > "gpurun_out/$(basename "$f" .py).log" 2>&1
    note "py $f $* rc=$?"; tail -15 "gpurun_out/$(basename "$f" .py).log" ;;
  seq)
    for p in "$@"; do eval "bash scripts/gpu_run.sh $p"; done ;;
  *) echo "unknown plan $plan"; exit 2 ;;
esac
