#!/bin/bash
# phase clocks of the board pass (instrumented library): gpurun -- bash scripts/gpu_r5_phases.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r34}; mkdir -p gpurun_out
timeout 300 python scripts/phase_timing_br.py 32768 4 > gpurun_out/${TAG}_br_phases.txt 2>&1
timeout 300 python scripts/phase_timing.py 32768 4 > gpurun_out/${TAG}_cfr_phases.txt 2>&1
cat gpurun_out/${TAG}_br_phases.txt gpurun_out/${TAG}_cfr_phases.txt | grep -v "^barrier\|^waiting"
