cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
(timeout 300 python tools/gbench.py gnn64 gnn512 gnn4096 --steps 20 > gpurun_out/g9_gbench.txt 2>&1); cat gpurun_out/g9_gbench.txt | tail -3
(timeout 300 python bench.py --workload schnet4096 --bf16 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/g9_schnet.json 2> gpurun_out/g9_schnet.err); cut -c1-200 gpurun_out/g9_schnet.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw; rocprofv3 --kernel-trace --stats -d /tmp/pw -o run -- python $GRAFT_REPO_ROOT/bench.py --workload schnet4096 --bf16 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/pw -name "*results.db" | head -1) 2>/dev/null | head -40 > $GRAFT_REPO_ROOT/gpurun_out/g9_schnet_stats.txt
