# End-of-round evidence on the GPU box in one call: GPU tests, smoke, counters of the workloads whose kernel sources changed
# (copied into profiles/ of the box's tree so that the bench line that follows reads records of the same sources), kernel
# stats of every bench workload, the bench lines.  Results under gpurun_out/ (copy gpurun_out/prof/<tag>_* and pmc_* and the
# bench files to profiles/ afterwards).      gpurun -- bash tools/final_round.sh r06 "lj108 lj4096 schnet4096rows"
TAG=${1:-r06}
PMC_WL=${2:-"lj108 lj4096 schnet4096rows"}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out gpurun_out/prof
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/f_pytest.log; tail -3 gpurun_out/f_pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > gpurun_out/f_smoke.log; tail -1 gpurun_out/f_smoke.log
bash tools/prof_round3.sh $TAG $PMC_WL > gpurun_out/f_prof.log 2>&1
for W in $PMC_WL; do cp gpurun_out/prof/pmc_$W.json profiles/; done
for W in lj108 schnet4096 water192 water192x64; do
  case " $PMC_WL " in *" $W "*) ;; *) bash tools/gpu_call.sh stats $TAG $W > /dev/null 2>&1 ;; esac
done
bash tools/prof_gnn_single.sh $TAG > /dev/null 2>&1
(python tools/kbench_cfconv.py --rows16 2>&1 | grep -v amdgpu.ids) > gpurun_out/prof/${TAG}_cfconv_kbench_stash.txt
(python tools/kbench_ring_mask.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/prof/${TAG}_ring_mask_kbench.txt
(timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err); tail -c 200 gpurun_out/f_bench.err
tail -c 600 gpurun_out/f_bench.json
