#!/bin/bash
# final check after the eltwise store policy: full GPU suite, eltwise bandwidth (the official table + by size)
OUT=gpurun_out/r3_ai; mkdir -p $OUT
timeout 800 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -2
python tools/eltwise_bw.py > $OUT/eltwise_bw.txt 2>/dev/null; cat $OUT/eltwise_bw.txt | cut -c1-110
python tools/sessions/elt_policy_sizes.py 2>/dev/null | tee $OUT/sizes.txt
