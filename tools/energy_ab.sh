#!/bin/bash
# Round 6: power and shader clock as tracked metrics.  Every variant loops for >= 5 s while
# tools/power_probe.py samples the card's hwmon node at 20 Hz (>= 100 samples after the warm-up): mean sclk, mean package
# power, and - from the variant's own us per launch - joules per launch.   tools/energy_ab.sh > profiles/r06_energy.txt
cd $GRAFT_REPO_ROOT
mkdir -p tools/bin
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -pragma-unroll-threshold=10000000 \
  -Ilagrangebench_amd/csrc -Iinclude tools/edge_ab.hip -o tools/bin/edge_ab 2> /tmp/edge_ab_build.log || { tail -5 /tmp/edge_ab_build.log; }
echo "# tools/energy_ab.sh: hwmon freq1_input / power1_input at 20 Hz while ONE variant loops (tools/bin/edge_ab E N iters 'name': 6 s of"
echo "# warm-up launches, then iters timed launches); joules per launch = mean power x us per launch"
for V in "compute only" "no GEMMs" "k_edge16v product (guard rows)" "k_edge16w<loads at top, defer nothing>" "k_edge16w<loads at top, defer all> last layer" "k_edge16v last layer"; do
  python tools/power_probe.py --hz 20 --skip 8 --label "edge_ab: $V" -- tools/bin/edge_ab 1037000 64000 24000 "$V" 2>&1 | grep -v "check\|differing\|^    E=\|INTERLEAVE" | awk '
    /\| n=/ { line=$0; match($0, /power +[0-9.]+ W/); pw=substr($0, RSTART+6, RLENGTH-8)+0 }
    / us  / { match($0, /[0-9.]+ us/); us=substr($0, RSTART, RLENGTH-3)+0; if (line != "") { printf "%s  | %.1f us/launch  %.4f J/launch\n", line, us, pw*us*1e-6; line="" } }'
done
echo "# the rollouts (bench.py; whole step, all kernels): J per step = mean power x ms per step"
python tools/power_probe.py --hz 20 --skip 6 --label "bench tgv3d x 8 GNS (stationary), 400-step rollouts" -- python bench.py --steps 400 --warmup 400 --repeats 4 --no-cpu-baseline --no-other-configs --no-f32 2>/dev/null | cut -c1-400
python tools/power_probe.py --hz 20 --skip 6 --label "bench dam2d x 8 SEGNN-10-64 (k_sg_msg 63 %), 1500-step rollouts" -- python bench.py --model segnn --workload dam2d --steps 1500 --warmup 1500 --no-cpu-baseline 2>/dev/null | cut -c1-400
