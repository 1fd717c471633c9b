#!/bin/bash
# (sample density rho, threshold rank) of the panel prefilter, now that rows short of candidates are proven from their
# buckets (k_bucket_rescore) instead of going to the exact kernel
for cfg in "100000 768 32" "1000000 384 16"; do
  for rho in 12 16 24 32; do for rank in 14 10 8 6; do
    echo -n "$cfg rho=$rho rank=$rank: "
    OSC_KNN_PANEL_RHO=$rho OSC_KNN_PANEL_RANK=$rank timeout -k 10 300 python scripts/knn_sym_ab.py $cfg 2>&1 | grep "sym=1" | sed 's/.*sym=1: //'
  done; done
done
