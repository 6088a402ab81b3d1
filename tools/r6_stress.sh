#!/bin/bash
# round 6: the hunt for the concurrent-handles corruption (NOTES/round5.md sections 8, 12) with tests/stress_handles.bin:
# which workloads fail side by side, and whether host threads, device co-residency or a kernel that scribbles LDS is what it takes
cd $GRAFT_REPO_ROOT
B=tests/stress_handles.bin
run() { echo "== $*"; ( timeout 240 "$@" 2>&1; echo "rc=$?" ) | grep -v amdgpu.ids | tail -${TAILN:-8}; }
IT=${IT:-30}
run $B --handles 5 --threads 5 --iters $IT --work mix --quiet
run $B --handles 5 --threads 1 --iters $IT --work mix --mode serial --quiet
run $B --handles 5 --iters $IT --work mix --mode interleave --quiet
run $B --handles 5 --iters $IT --work mix --mode serial --hog lds --quiet
run $B --handles 5 --iters $IT --work mix --mode serial --hog fill --quiet
for w in elem10 fem4 trip fd sum; do
  run $B --handles 4 --threads 4 --iters $IT --work $w --quiet
done
AMD_SERIALIZE_KERNEL=3 run $B --handles 5 --threads 5 --iters $IT --work mix --quiet
run $B --handles 5 --threads 5 --iters $IT --work mix --fresh 1 --quiet
