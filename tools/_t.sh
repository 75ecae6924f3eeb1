mkdir -p gpurun_out/r11a
timeout -k 10 800 python -m pytest tests/test_raster_gpu.py tests/test_train_ops_gpu.py -m gpu -x -q > gpurun_out/r11a/test.log 2>&1; echo "rc $?" >> gpurun_out/r11a/test.log; grep -v "^Extension modules" gpurun_out/r11a/test.log | tail -6
tools/ab_lib.sh r11a ab/lib_prev.so
