cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -20) > gpurun_out/g5_pytest.log
tail -12 gpurun_out/g5_pytest.log
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/g5_bench.json 2> gpurun_out/g5_bench.err); tail -c 300 gpurun_out/g5_bench.err
(timeout 300 python tools/gbench.py gnn64 gnn512 gnn4096 --steps 20 > gpurun_out/g5_gbench.txt 2>&1); cat gpurun_out/g5_gbench.txt | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw; rocprofv3 --kernel-trace --stats -d /tmp/pw -o run -- python $GRAFT_REPO_ROOT/bench.py --workload schnet4096 --bf16 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/pw -name "*results.db" | head -1) 2>/dev/null | head -60 > $GRAFT_REPO_ROOT/gpurun_out/g5_schnet_stats.txt
