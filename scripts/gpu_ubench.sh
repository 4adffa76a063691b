cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 ./scripts/ubench/lds_ubench > gpurun_out/lds_ubench.log 2>&1
cat gpurun_out/lds_ubench.log
bash scripts/gpu_ab.sh ${1:-f} ${2:-1}
