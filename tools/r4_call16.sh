# round 4, GPU call 16: GPU idle gaps inside the timed loop of the f32 / bf16-rows stacked passes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" "--bf16-rows"; do rm -rf /tmp/q0; rocprofv3 --kernel-trace --stats -d /tmp/q0 -o run -- python $R/bench.py --workload schnet4096 $v --steps 6 --warmup 3 --no-cpu-baseline > $R/$O/c16_bench$v.json 2>/dev/null
MS=$(python -c "
import json;d=json.load(open('$R/$O/c16_bench$v.json'));print(d['ms_per_step']*4)")
python -c "
import json;d=json.load(open('$R/$O/c16_bench$v.json'));print('schnet [$v]',d['value'],d['ms_per_step'])"
python $R/tools/rocpd_summary.py gaps $(find /tmp/q0 -name "*results.db" | head -1) 25 $MS > $R/$O/c16_gaps$v.txt 2>&1; head -64 $R/$O/c16_gaps$v.txt | cut -c1-170; done
