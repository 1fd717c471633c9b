set -u
R=$GRAFT_REPO_ROOT
cd $R
for CFG in "6500 768 32" "7000 512 16" "8192 384 16" "9000 256 16" "9000 1024 32" "14000 256 32" "14000 320 8" "8192 1536 32"; do
  for MR in "16384,32768" "4096,32768"; do echo "== $CFG OSC_XS_MIN_ROWS=$MR"; OSC_XS_MIN_ROWS=$MR timeout -k 10 200 python3 scripts/exp/settle_loop.py $CFG nochain 20 2>&1 | tail -1 | sed 's/.*settle_ms=/settle_ms=/' | cut -c1-18; done; done
