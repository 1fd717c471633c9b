#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r03_fi_tests.txt 2>&1 || { tail -40 $O/r03_fi_tests.txt; exit 1; }
tail -1 $O/r03_fi_tests.txt
for v in 1 2; do echo "## OSC_BLK_INIT=$v"; OSC_BLK_INIT=$v timeout -k 10 300 python scripts/config_times.py 2>&1 | grep "^c5\|^c3" | cut -c1-150; done
