#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r03_fi_tests.txt 2>&1 || { tail -40 $O/r03_fi_tests.txt; exit 1; }
tail -1 $O/r03_fi_tests.txt
for v in 1 2; do echo "## OSC_BLK_INIT=$v"; OSC_BLK_INIT=$v timeout -k 10 300 python scripts/shard_local_times.py c3 1 8 2>&1 | grep "^c3" | cut -c1-125; OSC_BLK_INIT=$v timeout -k 10 300 python scripts/config_times.py 2>&1 | cut -c1-150; done
