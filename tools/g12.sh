cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_verlet.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -40 > gpurun_out/g12_verlet.log
for sk in 0.02; do
MDG_VERLET_SKIN=$sk timeout 300 python bench.py --workload schnet4096 --bf16 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('skin', '$sk', d['value'], d['config'].get('neighbour_list',{}).get('searches_per_pass'))"
done
