# round 4, GPU call 26: large path after the prep / init changes: tests, counters of lj4096 re-collected, the bench line again
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out gpurun_out/prof; O=gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x -k "large or lj4096 or 4096_atoms or nve" 2>&1 | tail -6) > $O/c26_large.log; tail -3 $O/c26_large.log
bash tools/prof_round3.sh r04 lj4096 > $O/c26_prof.log 2>&1; cp $O/prof/pmc_lj4096.json profiles/; head -14 $O/prof/r04_lj4096_kernel_stats.txt | cut -c1-140
(MDG_BENCH_TRACE=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/f3_bench.json 2> $O/f3_bench.err); grep "trace lj4096" $O/f3_bench.err | cut -c1-300
python -c "
import json;d=json.load(open('$O/f3_bench.json'));ns=d['config']['north_star_workloads'];print(d['value'], ns['schnet4096']['value'], ns['schnet4096']['f32']['value'], ns['schnet4096']['bf16_rows']['value'], ns['lj4096']['value'], d.get('cpu_leg_errors'), d['secondary']['lj4096']['roofline'].get('counters','')[:80])"
(timeout 600 python bench.py --workload lj4096 --replicas 1 --steps 50 --warmup 8 --no-cpu-baseline > $O/c26_lj4096_r1.json 2>/dev/null); python -c "
import json;d=json.load(open('$O/c26_lj4096_r1.json'));print('lj4096 R=1',d['value'],d['ms_per_step'])"
