# round 4, GPU call 5: bf16 forward kernels (HASHD / BIASK), pipelined LJ adjoint sweep variants
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
(timeout 900 python -m pytest tests/test_gpu_fused_block.py tests/test_gpu_schnet.py tests/test_gpu_verlet.py tests/test_gpu_config5.py tests/test_gpu_secondary_pins.py tests/test_gpu_bonded.py -m gpu -q -x 2>&1 | tail -25) > $O/c5_schnet.log; tail -4 $O/c5_schnet.log
python tools/kbench_cfconv.py --bf16 > $O/c5_kbench_bf16.txt 2>&1; grep cfconv $O/c5_kbench_bf16.txt
(timeout 600 python bench.py --workload schnet4096 --bf16 --steps 12 --warmup 2 --no-cpu-baseline > $O/c5_bench_schnet_bf16.json 2> $O/c5_bench_schnet_bf16.err); python -c "
import json;d=json.load(open('$O/c5_bench_schnet_bf16.json'));print('schnet bf16',d['value'],d['ms_per_step'])"
for V in 1 2 3; do
  (MDG_LARGE_ADJ=$V timeout 300 python bench.py --workload lj4096 --steps 20 --warmup 3 --no-cpu-baseline > $O/c5_bench_lj4096_v$V.json 2>/dev/null); python -c "
import json;d=json.load(open('$O/c5_bench_lj4096_v$V.json'));print('lj4096 adj variant $V',d['value'],d['ms_per_step'])"
done
(timeout 900 python -m pytest tests/test_gpu_pins.py tests/test_gpu_secondary_pins.py tests/test_gpu_parity.py -m gpu -q -k "large or lj4096 or 4096_atoms or limits or timed_geometry" 2>&1 | tail -5) > $O/c5_large.log; tail -3 $O/c5_large.log
(MDG_LARGE_ADJ=3 timeout 900 python -m pytest tests/test_gpu_secondary_pins.py -m gpu -q -k "large" 2>&1 | tail -5) > $O/c5_large_v3.log; tail -2 $O/c5_large_v3.log
