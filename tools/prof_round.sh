# Evidence files of a round (run on the GPU box): prof_round.sh <tag>; results under gpurun_out/prof/, copy to profiles/
set -x
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/${TAG}_bench.json 2> /tmp/bench.err; tail -c 300 $O/${TAG}_bench.json
# kernel-time tables (rocprofv3 --kernel-trace --stats) of the three workloads
for W in lj108 schnet4096 lj4096; do LINES_SHOWN=3 bash $R/tools/prof_workload.sh $W ${TAG} > /dev/null 2>&1; done
LINES_SHOWN=3 bash $R/tools/prof_workload.sh schnet4096 ${TAG}bf16 --bf16 > /dev/null 2>&1      # (the bench line's secondary runs bf16 operands)
# HBM traffic of the headline kernels: FETCH_SIZE and WRITE_SIZE in separate passes
rm -rf /tmp/p2 /tmp/p3
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p2 -o run -- python $R/bench.py --workload lj108 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p3 -o run -- python $R/bench.py --workload lj108 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
F=$(find /tmp/p2 -name "*results.db" | head -1); W=$(find /tmp/p3 -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py pmc $F $W | grep -i "traj_\|rdf_\|^#\|^kernel" > $O/${TAG}_bench_pmc_fetch_write.txt
python $R/tools/pmc_to_json.py $F $W 16384 50 > $O/pmc_traffic.json
# issue-side counters
bash $R/tools/pmc_issue.sh ${TAG} > /dev/null 2>&1
bash $R/tools/pmc_cfconv.sh ${TAG} > /dev/null 2>&1
python $R/tools/kbench_cfconv.py > $O/${TAG}_cfconv_kbench.txt 2>/dev/null
python $R/tools/kbench_cfconv.py --bf16 > $O/${TAG}_cfconv_kbench_bf16.txt 2>/dev/null
ls -la $O | tail -20
cat $O/pmc_issue.json | head -30
