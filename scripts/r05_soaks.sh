#!/bin/bash
# Round-5 final soak pass on the GPU box (from the repo root): every differential soak once, tails into one file.
O=gpurun_out/r05_soaks.txt
echo "# round 5, final library: soak_cg_loop.py 6 40, soak_multirank.py 10 30, soak_panel.py 22 16, soak_sequences.py 5 30, soak_blocked_apply.py 3 24, soak_knn.py 5 16, soak_sharded_build.py 3 12, soak_streamed_create.py 41 16" > $O
run() { echo "== $1" >> $O; shift; timeout -k 10 500 "$@" > gpurun_out/soak_one.txt 2>&1; echo "rc=$?" >> $O; tail -3 gpurun_out/soak_one.txt | cut -c1-220 >> $O; }
run soak_cg_loop python tests/soak/soak_cg_loop.py 6 40
run soak_multirank python tests/soak/soak_multirank.py 10 30
run soak_panel python tests/soak/soak_panel.py 22 16
run soak_sequences python tests/soak/soak_sequences.py 5 30
run soak_blocked_apply python tests/soak/soak_blocked_apply.py 3 24
run soak_knn python tests/soak/soak_knn.py 5 16
run soak_sharded_build python tests/soak/soak_sharded_build.py 3 12
run soak_streamed_create python tests/soak/soak_streamed_create.py 41 16
grep -c "rc=0" $O
