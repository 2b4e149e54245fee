# issue-side counters of the headline kernels (two separate --pmc passes, kernel trace only); TAG = profile prefix
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/q1 /tmp/q2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d /tmp/q1 -o run -- python $R/bench.py --workload lj108 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/q2 -o run -- python $R/bench.py --workload lj108 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
A=$(find /tmp/q1 -name "*results.db" | head -1); B=$(find /tmp/q2 -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py pmc $A $B | grep -i "traj_\|rdf_\|^#\|^kernel" > $R/gpurun_out/prof/${TAG}_bench_pmc_issue.txt
python $R/tools/pmc_issue_json.py $A > $R/gpurun_out/prof/pmc_issue.json
cut -c1-150 $R/gpurun_out/prof/${TAG}_bench_pmc_issue.txt | grep -v "^#" | head -70
cat $R/gpurun_out/prof/pmc_issue.json
