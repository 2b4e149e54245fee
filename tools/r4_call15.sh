# round 4, GPU call 15: where the GPU idles in the f32 and bf16 stacked passes of bench.py (kernel trace, gaps)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" "--bf16-rows"; do rm -rf /tmp/q0; rocprofv3 --kernel-trace --stats -d /tmp/q0 -o run -- python $R/bench.py --workload schnet4096 $v --steps 4 --warmup 2 --no-cpu-baseline > $R/$O/c15_bench$v.json 2>/dev/null
python -c "
import json;d=json.load(open('$R/$O/c15_bench$v.json'));print('schnet [$v]',d['value'],d['ms_per_step'])"
python $R/tools/rocpd_summary.py gaps $(find /tmp/q0 -name "*results.db" | head -1) 30 > $R/$O/c15_gaps$v.txt 2>&1; head -70 $R/$O/c15_gaps$v.txt | cut -c1-170; done
