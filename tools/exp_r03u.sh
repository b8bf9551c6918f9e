#!/bin/bash
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 2400 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
