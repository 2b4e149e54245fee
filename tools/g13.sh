cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/g13_bench.json 2> gpurun_out/g13_bench.err); tail -c 300 gpurun_out/g13_bench.err
bash tools/prof_round3.sh r03b lj108 lj4096 schnet4096 > gpurun_out/g13_prof.log 2>&1
python tools/kbench_cfconv.py > gpurun_out/prof/r03b_cfconv_kbench.txt 2>/dev/null
python tools/kbench_cfconv.py --bf16 > gpurun_out/prof/r03b_cfconv_kbench_bf16.txt 2>/dev/null
tail -3 gpurun_out/prof/r03b_cfconv_kbench_bf16.txt
