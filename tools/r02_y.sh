#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "tile_pool" 2>&1 | tail -12
