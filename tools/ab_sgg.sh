#!/bin/bash
# General-irreps SEGNN forward A/B of two BUILDS of the library on one box: tools/ab_sgg.sh <tagA|product> <tagB|product>
cd $GRAFT_REPO_ROOT
L=lagrangebench_amd/csrc/liblbhip.so
cp $L /tmp/liblbhip_product.so
pick() { if [ "$1" = product ]; then cp /tmp/liblbhip_product.so $L; else cp tools/bin/var_$1/liblbhip.so $L; fi; }
for i in 1 2; do
  for V in $1 $2; do
    pick $V
    python tools/segnn_gen_bench.py --reps 20 2>&1 | grep general | sed -e "s/hidden .* path general//" -e "s/ \/ forward.*//" -e "s/^/$V /"
  done
done
pick product
