export PYTHONUNBUFFERED=1; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_stage1_gpu.py tests/test_punet_gpu.py tests/test_ops_gpu.py -v -x > gpurun_out/t_s1.log 2>&1; rc=$?; tail -5 gpurun_out/t_s1.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/bench_punet.py --detail 2>&1 | grep -v amdgpu | head -18
timeout -k 10 300 python3 tools/bench_stage1.py --steps 5 2>&1 | grep -v amdgpu | tail -3
bash tools/ab_step3.sh 2>&1 | tail -2
