#!/bin/bash
# bundle(10) under library variants: bundle_ab.sh NAME [NAME ...]  (oscillink_amd/liboscillink_hip_NAME.so), alternating twice
for rep in 1 2; do
for v in default "$@"; do
  if [ "$v" = default ]; then unset OSC_LIB_PATH; else export OSC_LIB_PATH=$PWD/oscillink_amd/liboscillink_hip_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python scripts/exp/r06/bundle_ab.py
done
done
