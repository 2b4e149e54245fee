cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_verlet.py tests/test_gpu_fused_block.py tests/test_gpu_config5.py tests/test_gpu_schnet.py -m gpu -x -q 2>&1 | tail -4
python tools/kbench_cfconv.py 2>/dev/null | head -4
python tools/kbench_cfconv.py --bf16 2>/dev/null | head -4
