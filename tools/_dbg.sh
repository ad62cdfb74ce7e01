python -m pytest tests/test_hip_sg2.py -x -q -m gpu -k "train_with" 2>&1 | grep -v "^  File\|^Extension" | head -20
