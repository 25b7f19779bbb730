#!/bin/bash
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -x -k "bf16 or c4 or fixture" 2>&1 | tail -4
echo "--- loader waves (default)"; python tools/sweep.py 2>/dev/null | grep "^bf16"
echo "--- no loader waves (TPP_HIP_BF16_LEGACY=2)"; TPP_HIP_BF16_LEGACY=2 python tools/sweep.py 2>/dev/null | grep "^bf16"
