# round 4, GPU call 12: host side after the caches (structure checks, layer parameters, filter descriptors, chain descriptors, raw stream)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out; R=$GRAFT_REPO_ROOT
python tools/hostprof_schnet.py --bf16-rows > $O/c12_hostprof.txt 2>&1; head -9 $O/c12_hostprof.txt | cut -c1-120
for v in bf16 bf16-rows bf16 bf16-rows; do (timeout 600 python bench.py --workload schnet4096 --$v --steps 12 --warmup 2 --no-cpu-baseline > $O/c12_bench_schnet_$v.json 2> $O/c12_bench_schnet_$v.err); python -c "
import json;d=json.load(open('$O/c12_bench_schnet_$v.json'));print('schnet $v',d['value'],d['ms_per_step'])"; done
(timeout 600 python bench.py --workload schnet4096 --steps 8 --warmup 2 --no-cpu-baseline > $O/c12_bench_schnet_f32.json 2> $O/c12_bench_schnet_f32.err); python -c "
import json;d=json.load(open('$O/c12_bench_schnet_f32.json'));print('schnet f32',d['value'],d['ms_per_step'],d['roofline']['step_roof']['frac'])"
(timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -30) > $O/c12_pytest.log; tail -5 $O/c12_pytest.log
