# rates of the 4 096-atom LJ liquid with stale lists (topology_update_freq = 5): fused (mdg_traj_*_large_stale) vs the generic
# path, and the fused rate at 64 stacked replicas beside topology_update_freq = 1
cd $GRAFT_REPO_ROOT
for A in "--freq 1" "--freq 5" "--freq 5 --generic" "--freq 1 --replicas 64" "--freq 5 --replicas 64" "--freq 2 --replicas 64"; do
  echo "== lj4096 $A"
  timeout 900 python tools/gbench.py lj4096 --steps 20 $A 2>&1 | grep -v amdgpu.ids | tail -1
done
