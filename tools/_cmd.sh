python -m pytest tests/test_gpu_sfno.py tests/test_gpu_bf16_storage.py tests/test_gpu_train_engine.py -q -x -m gpu 2>&1 | tail -3
