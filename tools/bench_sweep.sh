cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
run() { echo "== $*"; timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-pmc --no-f32 --repeats 1 --steps 10 --warmup 10 "$@" 2>/tmp/err.txt | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['value']/1e6,2), d['config'].get('workload'))
except Exception as e:
    print('FAILED', e)"; tail -2 /tmp/err.txt | grep -i "error\|Traceback" ; }
for W in tgv2d rpf2d ldc3d tgv3d dam2d; do for B in 1 2 8 16; do run --workload $W --batch $B; done; done
for W in tgv2d rpf2d ldc3d tgv3d dam2d; do for B in 1 4; do run --model segnn --workload $W --batch $B; done; done
run --workload tgv3d --batch 32
run --workload tgv3d --batch 8 --shuffle
run --workload tgv3d --batch 8 --mp-steps 5
