#!/bin/bash
# round 6: the fuzz's failing shapes in the native harness (three uneven parts of one ten-node mesh), warm and cold
cd $GRAFT_REPO_ROOT
B=tests/stress_handles.bin
run() { echo "== $*"; ( timeout 300 "$@" 2>&1; echo "rc=$?" ) | grep -v amdgpu.ids | tail -${TAILN:-8}; }
IT=${IT:-60}
run $B --handles 3 --threads 3 --iters $IT --work parts --kind 2 --mode spawn --quiet
HIP_LAUNCH_BLOCKING=1 run $B --handles 3 --threads 3 --iters $IT --work parts --kind 2 --mode spawn --quiet
HIP_LAUNCH_BLOCKING=1 run $B --handles 3 --threads 3 --iters $IT --work parts --kind 2 --mode spawn --fresh 1 --quiet
HIP_LAUNCH_BLOCKING=1 run $B --handles 6 --threads 6 --iters $IT --work parts --kind 2 --mode spawn --quiet
HIP_LAUNCH_BLOCKING=1 run $B --handles 3 --threads 3 --iters $IT --work parts --kind 2 --mode lockstep --quiet
$B --handles 3 --threads 3 --work parts --kind 2 --mode spawn --iters 1 --quiet --writeref gpurun_out/ref_parts.txt > /dev/null 2>&1
ok=0; bad=0
for i in $(seq 1 30); do
  if HIP_LAUNCH_BLOCKING=1 timeout 120 $B --handles 3 --threads 3 --work parts --kind 2 --mode spawn --iters 2 --quiet --cold 1 --ref gpurun_out/ref_parts.txt > gpurun_out/cold.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); echo "cold: FAIL $(grep -v amdgpu.ids gpurun_out/cold.log | head -3 | cut -c1-220)"; fi
done
echo "cold parts blocking: ok $ok fail $bad"
