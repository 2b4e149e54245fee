cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_verlet.py -m gpu -x -q 2>&1 | tail -5
for sk in 0.02 0.01 0.04; do
MDG_VERLET_SKIN=$sk timeout 300 python bench.py --workload schnet4096 --bf16 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('skin', '$sk', d['value'], d['config'].get('neighbour_list',{}).get('searches_per_pass'))"
done
