#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r03_lf_tests.txt 2>&1 || { tail -40 $O/r03_lf_tests.txt; exit 1; }
tail -1 $O/r03_lf_tests.txt
for v in 1 2; do echo "## OSC_X_DEFER=$v"; OSC_X_DEFER=$v timeout -k 10 300 python scripts/config_times.py 2>&1 | grep "^c[345]" | cut -c1-150;  OSC_X_DEFER=$v timeout -k 10 300 python scripts/mid_size_probe.py 2>&1 | grep "^N=" | cut -c1-60; done
