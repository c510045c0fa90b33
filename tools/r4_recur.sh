#!/bin/bash
# the recurrence kernel alone: one-stream forward kernel table + phase stamps of the diagnostic build
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 bench.py --no-cpu-baseline --no-train-step --no-secondary --streams 1 --windows 2 --kernel-table 2>&1 >/dev/null | grep -i "recur"
python3 -m pytest tests/test_gpu_parity.py -x -q -k "golden or stages" 2>&1 | tail -2
TRAJSDE_LIB=$PWD/trajsde_amd/variants/stamps.so python3 tools/phase_stamps.py recur 2>&1 | grep -v "^{" | tail -17
