# round 4, GPU call 3: f4 tests, packed + persistent listed kernels of the large path (pins + bench leg + kernel stats)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
(timeout 600 python -m pytest tests/test_gpu_bonded.py -m gpu -q 2>&1 | tail -25) > $O/c3_bonded.log; tail -4 $O/c3_bonded.log
(timeout 900 python -m pytest tests/test_gpu_pins.py tests/test_gpu_secondary_pins.py tests/test_gpu_parity.py -m gpu -q -k "large or lj or 4096 or limits or bonded" 2>&1 | tail -25) > $O/c3_large.log; tail -4 $O/c3_large.log
(timeout 600 python bench.py --workload lj4096 --steps 20 --warmup 3 > $O/c3_bench_lj4096.json 2> $O/c3_bench_lj4096.err); tail -c 300 $O/c3_bench_lj4096.err; python -c "
import json;d=json.load(open('$O/c3_bench_lj4096.json'));print('lj4096',d['value'],d['ms_per_step']);print(json.dumps(d.get('cpu_baseline'))[:1200])"
bash tools/prof_round3.sh r04b lj4096 > $O/c3_prof.log 2>&1; tail -14 $O/c3_prof.log | cut -c1-150
