#!/bin/bash
# round 5: the item partition's tests (generator in shuffled order, element batches, Base.sum over them) + config 4 without triplets
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "item or element or fem or lazy or flush_sum or odd_offsets" > gpurun_out/r5_items_pytest.log 2>&1; echo pytest_rc=$?; grep -E "passed|failed|Error|assert" gpurun_out/r5_items_pytest.log | tail -5
bash tools/r5_cfg4.sh 5
