#!/bin/bash
# round 6: the run-out-chain kernel at 4 / 5 / 6 waves per SIMD (library variants), kernel-trace summaries on one box.  gpurun -- bash scripts/gpu_r6h.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r86}; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
for V in "" _ctw4 _ctw6; do
  export POKERRL_AMD_LIB=$R/pokerrl_amd/lib/libpokerrl_hip$V.so
  echo "== variant [$V]"; bash scripts/gpu_r6e.sh ${TAG}$V | head -5 | cut -c1-180
done
