cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/g1_pytest.log
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/g1_bench.json 2> gpurun_out/g1_bench.err)
tail -c 600 gpurun_out/g1_bench.err
bash tools/prof_round3.sh r03a lj4096 schnet4096 lj108 > gpurun_out/g1_prof.log 2>&1
tail -5 gpurun_out/g1_pytest.log
