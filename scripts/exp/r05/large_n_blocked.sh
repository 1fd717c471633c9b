set -u
O=gpurun_out/r05ln; mkdir -p $O
for shape in "500000 384 16" "700000 384 16" "1000000 384 16" "600000 768 32"; do
  cfgs="OSC_SPMM_BLOCKED=-1"
  for g in 1 2 4; do for nb in 5 7 9 12; do cfgs="$cfgs OSC_SPMM_XS=1,OSC_XS_GROUPS=$g,OSC_SPMM_BLOCKED=$nb"; done; done
  timeout -k 10 600 python scripts/exp/blocked_apply/ab.py $shape $cfgs
done > $O/large_n.txt 2>&1
cut -c1-200 $O/large_n.txt
