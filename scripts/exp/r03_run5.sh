set -u
R=$GRAFT_REPO_ROOT
export OSC_LIB_PATH=$R/oscillink_amd/liboscillink_hip_nb48.so
timeout -k 10 900 python3 $R/scripts/exp/nb_sweep.py > $R/gpurun_out/r03_nb_shapes.txt 2>&1
cat $R/gpurun_out/r03_nb_shapes.txt
