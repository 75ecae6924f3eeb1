mkdir -p gpurun_out/r08e
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r08e/test_gpu.log 2>&1; echo "rc $?" >> gpurun_out/r08e/test_gpu.log
grep -v "^Extension modules" gpurun_out/r08e/test_gpu.log | tail -6
