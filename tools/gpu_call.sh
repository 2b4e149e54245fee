# One GPU call of a round (gpurun -- bash tools/gpu_call.sh <name> ...): small named recipes instead of one script per call.
#   tests <pytest -k expr>      GPU tests matching the expression (all when empty)
#   stats <tag> <workload>      rocprofv3 --kernel-trace --stats of bench.py --workload <workload> -> gpurun_out/prof/<tag>_<workload>_kernel_stats.txt
#   bench <outfile> [args]      python bench.py [args] -> gpurun_out/<outfile>
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
R=$GRAFT_REPO_ROOT
case "$1" in
  tests) shift; (timeout 1500 python -m pytest tests -m gpu -q -x ${1:+-k "$1"} 2>&1 | tail -15) ;;
  stats) TAG=$2; W=$3; shift 3; EXTRA="$@"; WW=$W
         [ "$W" = "schnet4096" ] && EXTRA="--bf16 $EXTRA"
         (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/q0 && rocprofv3 --kernel-trace --stats -d /tmp/q0 -o run -- python $R/bench.py --workload $WW --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null 2>&1
          DB0=$(find /tmp/q0 -name "*results.db" | head -1); python $R/tools/rocpd_summary.py stats $DB0 2>/dev/null | head -45 > $R/gpurun_out/prof/${TAG}_${W}_kernel_stats.txt)
         head -16 gpurun_out/prof/${TAG}_${W}_kernel_stats.txt | cut -c1-160 ;;
  bench) OUT=$2; shift 2; (timeout 1500 python bench.py "$@" > gpurun_out/$OUT 2> gpurun_out/$OUT.err); tail -c 400 gpurun_out/$OUT.err; tail -c 3000 gpurun_out/$OUT ;;
esac
