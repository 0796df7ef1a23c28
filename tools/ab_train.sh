#!/bin/bash
# Training-step A/B of two BUILDS of the library on one box: tools/ab_train.sh <tagA|product> <tagB|product>
# (variants from tools/build_variant.sh; "product" = the library in lagrangebench_amd/csrc).  Alternates A B A B.
cd $GRAFT_REPO_ROOT
A=$1; B=$2
L=lagrangebench_amd/csrc/liblbhip.so
cp $L /tmp/liblbhip_product.so
pick() { if [ "$1" = product ]; then cp /tmp/liblbhip_product.so $L; else cp tools/bin/var_$1/liblbhip.so $L; fi; }
for i in 1 2; do
  for V in $A $B; do
    pick $V
    for w in tgv2d tgv3d; do echo "$V $(python tools/train_profile.py $w 20 2>&1 | tail -1)"; done
    echo "$V $(python tools/train_profile.py dam2d 20 segnn 2>&1 | tail -1)"
  done
done
pick product
