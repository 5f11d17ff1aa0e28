timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 60 python tools/ab_lib.py || exit 1
timeout -k 10 60 python tools/ab_lib.py || exit 1
timeout -k 10 60 python tools/ab_lib.py 1280 720 || exit 1
