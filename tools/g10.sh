cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_verlet.py -m gpu -x -q 2>&1 | tail -25
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
(timeout 300 python tools/gbench.py gnn64 gnn512 gnn4096 --steps 20 > gpurun_out/g10_gbench.txt 2>&1); cat gpurun_out/g10_gbench.txt | tail -3
(timeout 300 python bench.py --workload schnet4096 --bf16 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/g10_schnet.json 2> gpurun_out/g10_schnet.err); cut -c1-200 gpurun_out/g10_schnet.json; tail -c 300 gpurun_out/g10_schnet.err
