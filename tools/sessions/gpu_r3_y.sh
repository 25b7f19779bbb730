#!/bin/bash
# fragment reads of the two-chunks-per-barrier tiles 3 k-steps ahead (rolling) against the second-half schedule (tools/_base = the
# library before the change), same box
OUT=gpurun_out/r3_y; mkdir -p $OUT
timeout 600 python -m pytest tests/test_chain_gpu.py tests/test_sharded_gpu.py tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -2
for i in 1 2; do echo "base"; LD_LIBRARY_PATH=tools/_base timeout 100 tools/mlp_probe --rows 256,512,1024 2>&1 | cut -c1-14,50-200; echo "new"; timeout 100 tools/mlp_probe --rows 256,512,1024 2>&1 | cut -c1-14,50-200; done > $OUT/ab.txt; cat $OUT/ab.txt
