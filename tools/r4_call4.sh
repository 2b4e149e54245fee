# round 4, GPU call 4: stale-list kernels (topology_update_freq > 1), listed kernels with all gathers up front
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "stale" 2>&1 | tail -30) > $O/c4_stale.log; tail -5 $O/c4_stale.log
(timeout 900 python -m pytest tests/test_gpu_pins.py tests/test_gpu_secondary_pins.py tests/test_gpu_parity.py -m gpu -q -k "large or lj4096 or 4096_atoms or limits or timed_geometry" 2>&1 | tail -25) > $O/c4_large.log; tail -4 $O/c4_large.log
(timeout 600 python bench.py --workload lj4096 --steps 20 --warmup 3 --no-cpu-baseline > $O/c4_bench_lj4096.json 2> $O/c4_bench_lj4096.err); tail -c 300 $O/c4_bench_lj4096.err; python -c "
import json;d=json.load(open('$O/c4_bench_lj4096.json'));print('lj4096',d['value'],d['ms_per_step'])"
(timeout 300 python bench.py --workload lj4096 --steps 20 --warmup 3 --no-cpu-baseline --replicas 1 > $O/c4_bench_lj4096_r1.json 2>/dev/null); python -c "
import json;d=json.load(open('$O/c4_bench_lj4096_r1.json'));print('lj4096 R=1',d['value'],d['ms_per_step'])"
bash tools/prof_round3.sh r04c lj4096 > $O/c4_prof.log 2>&1; tail -14 $O/c4_prof.log | cut -c1-150
