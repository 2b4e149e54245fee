cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_fused_block.py tests/test_gpu_schnet.py tests/test_gpu_config5.py tests/test_gpu_secondary_pins.py tests/test_gpu_pins.py -m gpu -x -q 2>&1 | tail -30) > gpurun_out/g3_pytest.log
tail -8 gpurun_out/g3_pytest.log
(timeout 300 python bench.py --workload schnet4096 --bf16 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/g3_schnet.json 2> gpurun_out/g3_schnet.err); tail -c 300 gpurun_out/g3_schnet.err; cut -c1-260 gpurun_out/g3_schnet.json
(timeout 300 python tools/gbench.py gnn64 gnn512 gnn4096 --steps 20 > gpurun_out/g3_gbench.txt 2>&1); cat gpurun_out/g3_gbench.txt | tail -5
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw; rocprofv3 --kernel-trace --stats -d /tmp/pw -o run -- python $GRAFT_REPO_ROOT/bench.py --workload schnet4096 --bf16 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/pw -name "*results.db" | head -1) 2>/dev/null | head -60 > $GRAFT_REPO_ROOT/gpurun_out/g3_schnet_stats.txt
