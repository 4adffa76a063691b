#!/bin/bash
# diagnostics: bench.py runs with the shader clock / power sampled beside them (is a slow run a low-clock run?)
# usage: scripts/gpu_clockwatch.sh tag n_runs [lib]
cd $GRAFT_REPO_ROOT; TAG=$1; N=${2:-4}; LIB=${3:-pokerrl_amd/lib/libpokerrl_hip.so}; mkdir -p gpurun_out
for i in $(seq 1 $N); do
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/${TAG}_clk_$i.txt 2>&1 &
  W=$!
  POKERRL_AMD_LIB=$PWD/$LIB python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('run $i: %.3f ms/iter kernel %.3f' % (j['ms_per_step'], j['roofline']['kernel_ms_per_iteration']))"
  kill $W; wait $W 2>/dev/null
  sort gpurun_out/${TAG}_clk_$i.txt | uniq -c | sort -rn | head -4
done
