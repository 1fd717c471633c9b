set -u
R=$GRAFT_REPO_ROOT
cd $R
for B in 0 4 6 8 9 10 12; do echo "== OSC_SPMM_BLOCKED=$B"; OSC_SPMM_BLOCKED=$B timeout -k 10 200 python scripts/shard_local_times.py c3 8 4 2>&1 | cut -c1-140; done
