TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/q0
rocprofv3 --kernel-trace --stats -d /tmp/q0 -o run -- python $R/tools/gbench.py gnn4096 --steps 20 --bf16-rows > /dev/null 2>&1
DB0=$(find /tmp/q0 -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py stats $DB0 2>/dev/null | head -60 > $O/${TAG}_gnn4096_single_kernel_stats.txt
cat $O/${TAG}_gnn4096_single_kernel_stats.txt | cut -c1-170
