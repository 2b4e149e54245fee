cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/g7_report.txt
MDG_TEST_REPORT=$GRAFT_REPO_ROOT/gpurun_out/g7_report.txt timeout 900 python -m pytest tests/test_gpu_config5.py tests/test_gpu_fused_block.py tests/test_gpu_torch_ops.py -m gpu -x -q 2>&1 | tail -12
(timeout 300 python bench.py --workload schnet4096 --bf16 --steps 8 --warmup 2 > gpurun_out/g7_schnet.json 2> gpurun_out/g7_schnet.err); cut -c1-200 gpurun_out/g7_schnet.json; tail -c 300 gpurun_out/g7_schnet.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw; rocprofv3 --kernel-trace --stats -d /tmp/pw -o run -- python $GRAFT_REPO_ROOT/bench.py --workload schnet4096 --bf16 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/pw -name "*results.db" | head -1) 2>/dev/null | head -30 > $GRAFT_REPO_ROOT/gpurun_out/g7_schnet_stats.txt
