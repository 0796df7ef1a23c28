#!/bin/bash
# Headline A/B of two BUILDS of the library on one box: tools/ab_lib.sh <tagA|product> <tagB|product> [bench args...]
# (variants from tools/build_variant.sh; "product" = the library in lagrangebench_amd/csrc).  Alternates A B A B.
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift; shift
L=lagrangebench_amd/csrc/liblbhip.so
cp $L /tmp/liblbhip_product.so
pick() { if [ "$1" = product ]; then cp /tmp/liblbhip_product.so $L; else cp tools/bin/var_$1/liblbhip.so $L; fi; }
for i in 1 2; do
  for V in $A $B; do
    pick $V
    python bench.py --no-cpu-baseline --no-other-configs --no-pmc --no-f32 --repeats 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', round(d['ms_per_step'],4), round(d['value']/1e6,2), d['breakdown_ms_per_step'])"
  done
done
pick product
