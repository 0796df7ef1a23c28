cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu-baseline --no-other-configs --no-pmc --no-f32 --repeats 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['breakdown_ms_per_step']; print(round(d['ms_per_step'],4), round(d['value']/1e6,2), 'cells', b.get('cells'), 'neighbors', b.get('neighbors'))"; }
for T in 0 1 0 1; do echo "LB_CELLS_TRAJ=$T tgv3d x8"; LB_CELLS_TRAJ=$T one; done
for T in 0 1; do echo "LB_CELLS_TRAJ=$T tgv2d x8"; LB_CELLS_TRAJ=$T one --workload tgv2d --batch 8; done
for T in 0 1; do echo "LB_CELLS_TRAJ=$T segnn dam2d x8"; LB_CELLS_TRAJ=$T one --model segnn --workload dam2d --batch 8; done
