#!/bin/bash
OUT=gpurun_out/${1:-r02e}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_sharded_gpu.py -m gpu -q -x -p no:cacheprovider -k "bf16 or sharded or fixture or random or c5 or c4" > $OUT/pytest_bf16.log 2>&1; echo "rc=$?" >> $OUT/pytest_bf16.log; tail -4 $OUT/pytest_bf16.log
timeout 120 python tools/abl_case_bf16.py "new" 2>/dev/null | grep "^bf16" | tee $OUT/bf16_new.txt
timeout 120 python tools/sweep.py bf16 2>/dev/null | grep "^bf16" | tee -a $OUT/bf16_new.txt
