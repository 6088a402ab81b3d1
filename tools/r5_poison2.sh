#!/bin/bash
# round 5 experiment: the whole gpu suite with released buffers poisoned and kept (see r5_poison.sh; the graveyard is capped at 64 GB)
export ESP_EXTRA_FLAGS=-DESP_EXPERIMENTS
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export ESP_POISON_FREE=1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/poison_pytest.log 2>&1; echo "rc=$?"; grep -E "^FAILED|^ERROR|passed|failed|POISON: [0-9]+ words|^E  " gpurun_out/poison_pytest.log | sort | uniq -c | cut -c1-220 | head -40
