python -m pytest tests/test_gpu_forward.py -q -x 2>&1 | grep -E "^E  |^>|passed|failed" | cut -c1-160 | head -8
