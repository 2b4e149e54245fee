# End-of-round evidence on the GPU box: tests, counters (copied into profiles/ of the box's tree so that the bench line that
# follows reads records of the same sources), the bench line, kernel benches.  Results under gpurun_out/ (copy the r0X_* /
# pmc_* files of gpurun_out/prof and the bench line to profiles/ afterwards).
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out gpurun_out/prof
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/f_pytest.log; tail -3 gpurun_out/f_pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > gpurun_out/f_smoke.log; tail -1 gpurun_out/f_smoke.log
bash tools/prof_round3.sh $TAG lj108 lj4096 schnet4096 schnet4096rows > gpurun_out/f_prof.log 2>&1
cp gpurun_out/prof/pmc_lj108.json gpurun_out/prof/pmc_lj4096.json gpurun_out/prof/pmc_schnet4096.json gpurun_out/prof/pmc_schnet4096rows.json profiles/
(timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err); tail -c 200 gpurun_out/f_bench.err
python tools/kbench_cfconv.py > gpurun_out/prof/${TAG}_cfconv_kbench.txt 2>/dev/null
python tools/kbench_cfconv.py --bf16 > gpurun_out/prof/${TAG}_cfconv_kbench_bf16.txt 2>/dev/null
python tools/kbench_cfconv.py --rows16 > gpurun_out/prof/${TAG}_cfconv_kbench_rows16.txt 2>/dev/null
(timeout 300 python tools/gbench.py gnn64 gnn512 gnn4096 --steps 20 > gpurun_out/f_gbench.txt 2>&1; timeout 300 python tools/gbench.py gnn64 gnn512 gnn4096 --steps 20 --bf16 >> gpurun_out/f_gbench.txt 2>&1; timeout 300 python tools/gbench.py gnn4096 --steps 10 --replicas 8 --bf16 >> gpurun_out/f_gbench.txt 2>&1); grep "steps/s" gpurun_out/f_gbench.txt
bash tools/prof_gnn_single.sh > /dev/null 2>&1
python tools/kbench_chain.py > gpurun_out/prof/${TAG}_chain_kbench.txt 2>/dev/null; python tools/kbench_gradjobs.py > gpurun_out/prof/${TAG}_gradjobs_kbench.txt 2>/dev/null
python tools/kbench_ring_mask.py > gpurun_out/prof/${TAG}_ring_mask_kbench.txt 2>/dev/null
python tools/hostprof_schnet.py --bf16-rows > gpurun_out/prof/${TAG}_hostprof_schnet_rows16.txt 2>/dev/null
python tools/hostprof_schnet.py --opt 12 > gpurun_out/prof/${TAG}_passes_schnet_bf16.txt 2>/dev/null
