cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/g2_pytest.log
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/g2_bench.json 2> gpurun_out/g2_bench.err)
tail -c 400 gpurun_out/g2_bench.err
tail -12 gpurun_out/g2_pytest.log
