set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python tests/soak/soak_blocked_apply.py 9 13 2>&1 | tail -8 | cut -c1-190
for CFG in "300000 768 32" "400000 512 32" "500000 384 16" "200000 1536 64 chain" "100000 768 32"; do echo "== $CFG"; timeout -k 10 200 python3 scripts/exp/settle_loop.py $CFG 4 2>&1 | tail -1 | sed 's/.*settle_ms=/settle_ms=/' | cut -c1-330; done
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "blocked or xcd or window or fullsize or config" 2>&1 | tail -3
