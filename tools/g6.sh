cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/g6_report.txt
MDG_TEST_REPORT=$GRAFT_REPO_ROOT/gpurun_out/g6_report.txt timeout 900 python -m pytest tests/test_gpu_config5.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
(timeout 300 python bench.py --workload schnet4096 --bf16 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/g6_schnet.json 2> gpurun_out/g6_schnet.err); cut -c1-200 gpurun_out/g6_schnet.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw; rocprofv3 --kernel-trace --stats -d /tmp/pw -o run -- python $GRAFT_REPO_ROOT/bench.py --workload schnet4096 --bf16 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/pw -name "*results.db" | head -1) 2>/dev/null | grep -i "scan_kernel\|launches"
