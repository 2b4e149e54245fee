cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_config5.py tests/test_gpu_fused_block.py tests/test_gpu_verlet.py tests/test_gpu_schnet.py -m gpu -x -q 2>&1 | tail -2
python tools/kbench_cfconv.py 2>/dev/null | head -3
python tools/kbench_cfconv.py --bf16 2>/dev/null | head -3
timeout 300 python bench.py --workload schnet4096 --bf16 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-160
