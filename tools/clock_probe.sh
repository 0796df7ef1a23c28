#!/bin/bash
# Shader / memory clocks and socket power while ONE variant of tools/edge_ab runs for a few seconds (rocm-smi sampled twice a
# second): is the edge kernel's "compute + memory do not overlap" a power / clock effect?   tools/clock_probe.sh > out.txt
cd $GRAFT_REPO_ROOT
for V in "compute only" "no GEMMs" "k_edge16v product (guard rows)" "k_edge16w<loads at top, defer nothing>" "k_edge16v last layer"; do
  echo "== $V"
  tools/bin/edge_ab 1037000 64000 20000 "$V" > /tmp/ab_$$.txt 2>&1 &
  P=$!
  sleep 9
  for i in 1 2 3 4 5 6; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' ';'
    echo
    sleep 0.5
  done
  wait $P
  grep -E "us  " /tmp/ab_$$.txt | head -3
done
