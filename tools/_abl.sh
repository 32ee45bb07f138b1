timeout 900 python -m pytest tests/test_hip_model.py -m gpu -x -q -k "pretrained_like" 2>&1 | tail -12
