cd $GRAFT_REPO_ROOT
sed -i 's#/root/repo#'$GRAFT_REPO_ROOT'#g' tools/tbl_err.py
timeout 600 python tools/tbl_err.py 2>&1 | grep -v amdgpu.ids | tail -6
for NODES in 0 2048; do
  echo "== mlp108 table nodes $NODES (0 = default)"
  timeout 600 python tools/gbench.py mlp108 --replicas 8192 --steps 49 --table-nodes $NODES 2>&1 | grep -v amdgpu.ids | tail -1
done
