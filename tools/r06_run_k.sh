timeout 900 python -m pytest tests/test_gpu_batch_large.py -m gpu -x -q 2>&1 | tail -15
