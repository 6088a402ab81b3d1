#!/bin/bash
# round 4: the host boundary (esp_append_host packing + transfers, esp_get_csc), a few repetitions in fresh processes (run through gpurun)
cd $GRAFT_REPO_ROOT
nproc; lscpu | grep -E "Model name|NUMA node\(s\)|Thread" | head -4
for rep in 1 2 3; do
  ESP_HOST_TRACE=1 ESP_EXTRA_ONLY=cfg2 python3 tools/r4_extra.py 2 2>&1 | grep -E "esp_append_host:|cfg2_host" | tail -2 | cut -c1-160,250-520
done
