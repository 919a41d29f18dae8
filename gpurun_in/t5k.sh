cd $GRAFT_REPO_ROOT
time python -m pytest tests/test_full_size.py -m gpu -x -q -k "hot_path_5000" 2>&1 | tail -25
