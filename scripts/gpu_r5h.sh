#!/bin/bash
( time timeout 400 python -m pytest tests/test_hetero_sample_gpu.py tests/test_sample_gpu.py -m gpu -x -q ) 2>&1 | tail -40
