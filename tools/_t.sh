mkdir -p gpurun_out/r09b
timeout -k 10 800 python -m pytest tests/test_raster_gpu.py -m gpu -x -q > gpurun_out/r09b/test.log 2>&1; echo "rc $?" >> gpurun_out/r09b/test.log; grep -v "^Extension modules" gpurun_out/r09b/test.log | tail -6
tools/ab_lib.sh r09b ab/lib_prev.so
