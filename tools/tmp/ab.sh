timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 100 python tools/lbvh_time.py terrain grid grid4k || exit 1
