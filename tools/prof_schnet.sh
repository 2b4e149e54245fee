# kernel-time table of the 4096-bead x 8 SchNet workload (bench.py --workload schnet4096); TAG = profile prefix
set -x
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --workload schnet4096 --steps 5 --warmup 2 > $R/gpurun_out/prof/${TAG}_bench_schnet4096.json 2>/tmp/bs.err; tail -c 1500 $R/gpurun_out/prof/${TAG}_bench_schnet4096.json
rm -rf /tmp/p4
rocprofv3 --kernel-trace --stats -d /tmp/p4 -o run -- python $R/bench.py --workload schnet4096 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/p4 -name "*results.db" | head -1) | head -45 > $R/gpurun_out/prof/${TAG}_schnet4096x8_kernel_stats.txt
head -30 $R/gpurun_out/prof/${TAG}_schnet4096x8_kernel_stats.txt | cut -c1-150
