cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py tests/test_hip_sg3.py -x -q -m gpu -k "filtered or flr or sg3 or network" 2>&1 | tail -5
for cfg in "11 fwd" "10 fwd" "8 fwd" "12 fwd"; do echo "$(python tools/flr_one.py $cfg 30 16 2>&1 | tail -1)"; done
