cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
(MDG_BENCH_TRACE=1 timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/c21_bench.json 2> $O/c21_bench.err); grep "trace" $O/c21_bench.err | cut -c1-700
python -c "
import json;d=json.load(open('$O/c21_bench.json'));ns=d['config']['north_star_workloads'];print(d['value'], ns['schnet4096']['value'], ns['schnet4096']['f32']['value'], ns['schnet4096']['bf16_rows']['value'], ns['lj4096']['value'])"
