#!/bin/bash
# call 44: the pair tests with the default policy's test
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_pair_search.py -m gpu -x -q 2>&1 | tail -8
