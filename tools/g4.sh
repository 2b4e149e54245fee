cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_fused_block.py tests/test_gpu_schnet.py tests/test_gpu_config5.py -m gpu -x -q --durations=8 2>&1 | tail -40) > gpurun_out/g4_pytest.log
tail -25 gpurun_out/g4_pytest.log
