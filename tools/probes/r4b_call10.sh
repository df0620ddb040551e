mkdir -p gpurun_out
python -m pytest tests/test_gpu_models.py -q 2>&1 | grep -v Warn | tail -40
