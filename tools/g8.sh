cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw; rocprofv3 --kernel-trace --stats -d /tmp/pw -o run -- python $GRAFT_REPO_ROOT/tools/gbench.py gnn512 --steps 20 > /tmp/gb.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/pw -name "*results.db" | head -1) 2>/dev/null | head -70 > $GRAFT_REPO_ROOT/gpurun_out/g8_gnn512_stats.txt
tail -2 /tmp/gb.log
