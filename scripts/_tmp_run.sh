set -e
timeout -k 10 600 python -m pytest tests/test_search_gpu.py -x -q 2>&1 | tail -5
