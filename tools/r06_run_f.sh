mkdir -p gpurun_out/r06_f
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06_f/pytest.log 2>&1; tail -3 gpurun_out/r06_f/pytest.log
