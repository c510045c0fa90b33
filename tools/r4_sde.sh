#!/bin/bash
# SDE step: parity tests that touch it, the step bench at both sizes, phase stamps of the diagnostic build
python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
python3 tools/sde_step_bench.py 786432 | cut -c1-170
python3 tools/sde_step_bench.py 49152 | cut -c1-170
TRAJSDE_LIB=$PWD/trajsde_amd/variants/stamps.so python3 tools/phase_stamps.py sde_step 2>&1 | grep -v "^{" | tail -9
