#!/bin/bash
# phase shares (wave 0 shader clocks, light instrumentation) of the best-response pass and of the CFR+ update passes
cd $GRAFT_REPO_ROOT; TAG=${1:-r4b}; mkdir -p gpurun_out
timeout 300 python scripts/phase_timing_br.py 32768 4 > gpurun_out/${TAG}_br_phases.txt 2>&1; cat gpurun_out/${TAG}_br_phases.txt | head -12
timeout 300 python scripts/phase_timing.py 32768 4 > gpurun_out/${TAG}_cfr_phases.txt 2>&1; cat gpurun_out/${TAG}_cfr_phases.txt | head -12
