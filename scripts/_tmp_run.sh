set -e
timeout -k 10 600 python -m pytest tests/test_search_gpu.py -x -q -k "tech or rrf or large_k or random" 2>&1 | tail -3
python scripts/hybrid_bench.py
