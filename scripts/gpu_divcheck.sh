# scripts/ubench/div_check.hip on the GPU box: shared-reciprocal packed regret matching vs the generic division -> gpurun_out/div_check.log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result -Ipokerrl_amd/csrc -Iinclude scripts/ubench/div_check.hip -o /tmp/div_check
timeout 300 /tmp/div_check > gpurun_out/div_check.log 2>&1; echo "rc=$?" >> gpurun_out/div_check.log
cat gpurun_out/div_check.log
