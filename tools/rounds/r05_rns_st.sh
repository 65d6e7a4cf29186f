#!/bin/bash
# The int8 residue product experiment (tools/probes/rns_product): ceiling, correctness, timing, knock-outs.
# Build first (here or on the box): bash tools/probes/rns_product/build.sh
D=tools/probes/rns_product
O=gpurun_out/r05j; mkdir -p $O
timeout 120 $D/ceiling > $O/int8_residue_ceiling.txt 2>&1
{
timeout 100 $D/check 300 100 70 3
timeout 100 $D/check 1000 1297 600 6
RNS_NBUF=3 timeout 100 $D/check 1000 1297 600 6
timeout 300 $D/check 100405 1297 286 6
timeout 300 $D/check 100405 1297 286 0
timeout 300 $D/check 100405 1297 286 12
for nb in 2 3; do for kn in 0 1 2 4 3 7; do echo "ring of $nb buffers, knock-out $kn (1: no fold, 2: no products, 4: no loads after the first stages; results wrong, timing only)"; RNS_QUICK=1 RNS_NBUF=$nb RNS_KNOCK=$kn timeout 200 $D/check 100405 1297 286 6 | sed 's/.*| residues/residues/'; done; done
for st in 3 4 5 6; do echo "XCD super-tile of 2^$st row tiles x 2^(6-$st) column tiles"; RNS_QUICK=1 RNS_ST=$st timeout 200 $D/check 100405 1297 286 6 | sed 's/.*| residues/residues/'; done
} > $O/int8_residue_product.txt 2>&1
tail -3 $O/int8_residue_product.txt
