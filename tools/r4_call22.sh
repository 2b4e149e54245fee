cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
(MDG_BENCH_TRACE=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/f2_bench.json 2> $O/f2_bench.err); grep "trace" $O/f2_bench.err | cut -c1-400
python -c "
import json;d=json.load(open('$O/f2_bench.json'));ns=d['config']['north_star_workloads'];print(d['value'], ns['schnet4096']['value'], ns['schnet4096']['f32']['value'], ns['schnet4096']['bf16_rows']['value'], ns['lj4096']['value'], d.get('cpu_leg_errors'), d['cpu_baseline']['value'], ns['schnet4096']['cpu_baseline_value'], ns['lj4096']['cpu_baseline_value'])"
