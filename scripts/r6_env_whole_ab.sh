mkdir -p gpurun_out/r114
for rep in 1 2 3; do for w in 0 1; do
if [ $w = 0 ]; then export PRL_EB_NO_WHOLE=1; else unset PRL_EB_NO_WHOLE; fi
echo -n "whole-chunk kernel $w: "; POKERRL_AMD_LIB=$PWD/pokerrl_amd/lib/libpokerrl_hip_ebw.so python bench_env.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms frac %.3f hands %d pot %.6f' % (d['ms_per_step'], d['roofline']['frac'], d['config']['hands_finished'], d['config']['mean_pot']))"
done; done | tee gpurun_out/r114/whole_ab.txt
unset PRL_EB_NO_WHOLE
POKERRL_AMD_LIB=$PWD/pokerrl_amd/lib/libpokerrl_hip_ebw.so timeout 600 python -m pytest tests/test_envbatch.py -m gpu -x -q 2>&1 | tail -2
