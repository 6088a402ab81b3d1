#!/bin/bash
# round 6: flushes that START together (lockstep / spawn), per workload
cd $GRAFT_REPO_ROOT
B=tests/stress_handles.bin
run() { echo "== $*"; ( timeout 240 "$@" 2>&1; echo "rc=$?" ) | grep -v amdgpu.ids | tail -${TAILN:-8}; }
IT=${IT:-40}
for m in lockstep spawn; do
  for w in elem10 fem4 trip fd mix; do
    run $B --handles 4 --threads 4 --iters $IT --work $w --mode $m --quiet
  done
done
run $B --handles 8 --threads 8 --iters $IT --work elem10 --mode lockstep --quiet
run $B --handles 3 --threads 3 --iters $IT --work elem10 --mode spawn --fresh 1 --quiet
