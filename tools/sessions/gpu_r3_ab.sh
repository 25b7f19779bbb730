#!/bin/bash
# chain kernel without the run-time stamp hooks (shipped) against the library before (tools/_base), same box; stamps from the side build
OUT=gpurun_out/r3_ab; mkdir -p $OUT
timeout 300 python -m pytest tests/test_chain_gpu.py tests/test_sharded_gpu.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -2
for i in 1 2; do echo "base"; LD_LIBRARY_PATH=tools/_base timeout 100 tools/mlp_probe 2>&1 | cut -c1-14,50-200; echo "new"; timeout 100 tools/mlp_probe 2>&1 | cut -c1-14,50-200; done > $OUT/ab.txt; cat $OUT/ab.txt
LD_LIBRARY_PATH=tools/_abl TPP_HIP_CHAIN_STAMPS=$OUT/stamps_512.txt timeout 60 tools/mlp_probe --only chain --rows 512 --iters 50 > /dev/null 2>&1; python tools/stamps_report.py $OUT/stamps_512.txt | head -3
