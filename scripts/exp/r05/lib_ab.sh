#!/bin/bash
# settle / apply times of config 3 (and a second shape) under variant builds of the library: lib_ab.sh NAME [NAME ...]
for v in default "$@"; do
  if [ "$v" = default ]; then unset OSC_LIB_PATH; else export OSC_LIB_PATH=$PWD/oscillink_amd/liboscillink_hip_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python scripts/exp/r05/blk_variant_ab.py 100000 768 32 -1
  timeout -k 10 200 python scripts/exp/r05/blk_variant_ab.py 200000 1536 64 -1
done
