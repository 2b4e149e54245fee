set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/prof/r01h_bench.json 2> /tmp/bench.err; tail -1 $R/gpurun_out/prof/r01h_bench.json | cut -c1-200
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o run -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/p1 -name "*results.db" | head -1) > $R/gpurun_out/prof/r01h_bench_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p2 -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p3 -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
F=$(find /tmp/p2 -name "*results.db" | head -1); W=$(find /tmp/p3 -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py pmc $F $W > $R/gpurun_out/prof/r01h_bench_pmc_fetch_write.txt
python $R/tools/pmc_to_json.py $F $W 16384 50 > $R/gpurun_out/prof/pmc_traffic.json
python $R/bench.py --workload schnet4096 --steps 3 --warmup 1 > $R/gpurun_out/prof/r01h_bench_schnet4096.json 2>/dev/null; tail -1 $R/gpurun_out/prof/r01h_bench_schnet4096.json | cut -c1-200
rocprofv3 --kernel-trace --stats -d /tmp/p4 -o run -- python $R/bench.py --workload schnet4096 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/p4 -name "*results.db" | head -1) | head -40 > $R/gpurun_out/prof/r01h_schnet4096x8_kernel_stats.txt
ls -la $R/gpurun_out/prof
bash $R/tools/pmc_issue.sh > /dev/null 2>&1
cp $R/gpurun_out/prof/pmc_issue.txt $R/gpurun_out/prof/r01h_bench_pmc_issue.txt
