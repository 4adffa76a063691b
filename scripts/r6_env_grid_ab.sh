mkdir -p gpurun_out/r105
for rep in 1 2; do for cap in 4096 1024 2048 1280; do
echo -n "cap $cap: "; PRL_EB_GRID_CAP=$cap POKERRL_AMD_LIB=$PWD/pokerrl_amd/lib/libpokerrl_hip_ebr6.so python bench_env.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
done; done | tee gpurun_out/r105/grid_ab.txt
