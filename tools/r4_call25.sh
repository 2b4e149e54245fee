# round 4, GPU call 25: large_prep with one round trip per atom (rows loaded together): tests + lj4096 rates (64 replicas, 1 replica)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out; R=$GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q -x -k "large or lj4096 or 4096_atoms or stacked_large" 2>&1 | tail -6) > $O/c25_large.log; tail -3 $O/c25_large.log
(timeout 600 python bench.py --workload lj4096 --steps 50 --warmup 8 --no-cpu-baseline > $O/c25_lj4096.json 2>/dev/null); python -c "
import json;d=json.load(open('$O/c25_lj4096.json'));print('lj4096',d['value'],d['ms_per_step'])"
(timeout 600 python bench.py --workload lj4096 --replicas 1 --steps 50 --warmup 8 --no-cpu-baseline > $O/c25_lj4096_r1.json 2>/dev/null); python -c "
import json;d=json.load(open('$O/c25_lj4096_r1.json'));print('lj4096 R=1',d['value'],d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/q0; rocprofv3 --kernel-trace --stats -d /tmp/q0 -o run -- python $R/bench.py --workload lj4096 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/q0 -name "*results.db" | head -1) 2>/dev/null | head -12 | cut -c1-140
