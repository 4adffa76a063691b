# LDS / cross-lane micro-benchmarks (scripts/ubench/lds_ubench.hip, built here with hipcc) -> gpurun_out/lds_ubench.log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/lds_ubench.hip -o scripts/ubench/lds_ubench
timeout 300 ./scripts/ubench/lds_ubench > gpurun_out/lds_ubench.log 2>&1
cat gpurun_out/lds_ubench.log
