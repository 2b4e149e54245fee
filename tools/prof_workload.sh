# kernel-time table of one bench workload: prof_workload.sh <workload> <tag> [extra bench args]
W=$1; TAG=$2; shift 2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw
rocprofv3 --kernel-trace --stats -d /tmp/pw -o run -- python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/pw -name "*results.db" | head -1) 2>/dev/null | head -40 > $R/gpurun_out/prof/${TAG}_${W}_kernel_stats.txt
head -${LINES_SHOWN:-16} $R/gpurun_out/prof/${TAG}_${W}_kernel_stats.txt | cut -c1-175
