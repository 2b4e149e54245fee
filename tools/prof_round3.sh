# Counter evidence of a round for the three bench workloads (run on the GPU box): prof_round3.sh <tag> [workloads...]
# Per workload: one --kernel-trace --stats pass and separate --pmc passes (FETCH_SIZE | WRITE_SIZE | two issue groups);
# results: gpurun_out/prof/<tag>_<workload>_kernel_stats.txt and gpurun_out/prof/pmc_<workload>.json (copy to profiles/).
TAG=${1:-r03}; shift
WL=${@:-lj108 lj4096 schnet4096}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in $WL; do
  EXTRA=""; WW=$W; [ "$W" = "schnet4096" ] && EXTRA="--bf16"
  [ "$W" = "schnet4096rows" ] && { EXTRA="--bf16-rows"; WW=schnet4096; }      # (the rows16 precision option: records of its own)
  BENCH="python $R/bench.py --workload $WW --steps 2 --warmup 1 --no-cpu-baseline $EXTRA"
  rm -rf /tmp/q0 /tmp/q1 /tmp/q2 /tmp/q3 /tmp/q4
  rocprofv3 --kernel-trace --stats -d /tmp/q0 -o run -- $BENCH > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/q1 -o run -- $BENCH > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/q2 -o run -- $BENCH > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace -d /tmp/q3 -o run -- $BENCH > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d /tmp/q4 -o run -- $BENCH > /dev/null 2>&1
  DB0=$(find /tmp/q0 -name "*results.db" | head -1)
  python $R/tools/rocpd_summary.py stats $DB0 2>/dev/null | head -45 > $O/${TAG}_${W}_kernel_stats.txt
  python $R/tools/pmc_collect.py $W $DB0 $(find /tmp/q1 /tmp/q2 /tmp/q3 /tmp/q4 -name "*results.db") > $O/pmc_${W}.json 2> $O/pmc_${W}.err
  head -14 $O/${TAG}_${W}_kernel_stats.txt | cut -c1-150
done
ls -la $O | tail
