#!/bin/bash
# sample density of the panel prefilter's thresholds after the half sweep / coarse entries (round 3's optimum was rho = 12)
for cfg in "100000 768 32" "1000000 384 16" "200000 768 64"; do
  for rho in 10 12 14 16 20 24; do
    echo -n "$cfg rho=$rho: "
    OSC_KNN_PANEL_RHO=$rho timeout -k 10 300 python scripts/knn_sym_ab.py $cfg 2>&1 | grep "sym=1" | sed 's/.*sym=1: //'
  done
done
