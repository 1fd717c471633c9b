#!/bin/bash
# Round-6 measurement pass (run on the GPU box from the repo root): everything DESIGN.md quotes from profiles/r06_*.
set -u
O=gpurun_out/r06m; mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 600 python scripts/config_times.py > $O/config_times.txt 2>&1
timeout -k 10 300 python scripts/request_latency.py > $O/request_latency.txt 2>&1
{ timeout -k 10 400 python scripts/shard_local_times.py c3 1 2 4 8; timeout -k 10 400 python scripts/shard_local_times.py c5 1 4; timeout -k 10 500 python scripts/shard_local_times.py c4 1 8; } > $O/shard_local_times.txt 2>&1
timeout -k 10 300 python scripts/mid_size_probe.py > $O/mid_size_probe.txt 2>&1
{ timeout -k 10 300 python scripts/exp/r05/blk_stamps.py 100000 768 32 6; timeout -k 10 300 python scripts/exp/r05/blk_stamps.py 200000 1536 64 6; } > $O/blk_stamps.txt 2>&1
{ for c in "100000 768 32" "1000000 384 16"; do OSC_AB_SYM_ONLY=1 timeout -k 10 300 python scripts/knn_sym_ab.py $c; done; timeout -k 10 300 python scripts/knn_only.py 200000 1536 64; } > $O/knn_build.txt 2>&1
{ timeout -k 10 200 python scripts/exp/r05/sharded_build_times.py 100000 768 32 8; timeout -k 10 300 python scripts/exp/r05/sharded_build_times.py 200000 1536 64 4; timeout -k 10 400 python scripts/exp/r05/sharded_build_times.py 1000000 384 16 8; } > $O/sharded_build_times.txt 2>&1
tail -3 $O/config_times.txt | cut -c1-200
