mkdir -p gpurun_out/r11c
timeout -k 10 800 python -m pytest tests/test_raster_gpu.py -m gpu -x -q > gpurun_out/r11c/test.log 2>&1; echo "rc $?" >> gpurun_out/r11c/test.log; grep -v "^Extension modules" gpurun_out/r11c/test.log | tail -4
tools/ab_lib.sh r11c ab/lib_prev.so
