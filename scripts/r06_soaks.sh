#!/bin/bash
# Round-6 soak pass on the GPU box (from the repo root): the differential soaks that touch the lattice build (the round's kernel
# work: hand-placed K loop, two row groups at D <= 768, pair-form entries), with the planner's choice and with two row groups
# forced at every size; the CG soaks once.  Tails into one file.
S=${1:-0}   # seed offset: a second pass with fresh seeds is `bash scripts/r06_soaks.sh 100`
O=gpurun_out/r06_soaks.txt
mkdir -p gpurun_out
echo "# seed offset $S" > $O; echo "# round 6, final library: soak_panel.py 31 20 (planner) / 32 20 (OSC_KNN_PANEL_NRG=2) / 33 12 (OSC_KNN_PANEL_NRG=2 OSC_KNN_PANEL_T=6), soak_knn.py 6 16, soak_sharded_build.py 4 12 (planner / NRG=2), soak_streamed_create.py 42 16 (planner / NRG=2), soak_cg_loop.py 7 30, soak_blocked_apply.py 4 16, soak_sequences.py 6 20, soak_multirank.py 11 20" >> $O
run() { echo "== $1" >> $O; shift; timeout -k 10 500 "$@" > gpurun_out/soak_one.txt 2>&1; echo "rc=$?" >> $O; tail -3 gpurun_out/soak_one.txt | cut -c1-220 >> $O; }
run soak_panel python tests/soak/soak_panel.py $((31+S)) 20
export OSC_KNN_PANEL_NRG=2
run soak_panel_nrg2 python tests/soak/soak_panel.py $((32+S)) 20
OSC_KNN_PANEL_T=6 run soak_panel_nrg2_T6 python tests/soak/soak_panel.py $((33+S)) 12
run soak_sharded_build_nrg2 python tests/soak/soak_sharded_build.py $((5+S)) 8
run soak_streamed_create_nrg2 python tests/soak/soak_streamed_create.py $((43+S)) 10
unset OSC_KNN_PANEL_NRG
run soak_knn python tests/soak/soak_knn.py $((6+S)) 16
run soak_sharded_build python tests/soak/soak_sharded_build.py $((4+S)) 12
run soak_streamed_create python tests/soak/soak_streamed_create.py $((42+S)) 16
run soak_cg_loop python tests/soak/soak_cg_loop.py $((7+S)) 30
run soak_blocked_apply python tests/soak/soak_blocked_apply.py $((4+S)) 16
run soak_sequences python tests/soak/soak_sequences.py $((6+S)) $((20+S))  # (first seed, one past the last)
run soak_multirank python tests/soak/soak_multirank.py $((11+S)) 20
grep -c "rc=0" $O
