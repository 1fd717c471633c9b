#!/bin/bash
# Round-4 measurement pass (run on the GPU box from the repo root): everything DESIGN.md quotes from profiles/r04_*.
set -u
O=gpurun_out/r04m; mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 600 python scripts/config_times.py > $O/config_times.txt 2>&1
timeout -k 10 300 python scripts/request_latency.py > $O/request_latency.txt 2>&1
{ timeout -k 10 400 python scripts/shard_local_times.py c3 1 2 4 8; timeout -k 10 400 python scripts/shard_local_times.py c5 1 4; timeout -k 10 500 python scripts/shard_local_times.py c4 1 8; } > $O/shard_local_times.txt 2>&1
timeout -k 10 300 python scripts/mid_size_probe.py > $O/mid_size_probe.txt 2>&1
timeout -k 10 300 python scripts/transfer_times.py > $O/transfer_times.txt 2>&1
{ for c in "20000 600 32" "100000 768 32" "200000 384 16" "1000000 384 16"; do timeout -k 10 300 python scripts/knn_sym_ab.py $c; done; timeout -k 10 300 python scripts/knn_only.py 200000 1536 64; OSC_KNN_MODE=prefilter timeout -k 10 300 python scripts/knn_only.py 200000 1536 64; } > $O/knn_sym_ab.txt 2>&1
timeout -k 10 900 python scripts/shape_sweep.py > $O/shape_sweep.txt 2>&1
tail -3 $O/shape_sweep.txt | cut -c1-160
