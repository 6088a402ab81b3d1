#!/bin/bash
# round 6: the native harness under HIP_LAUNCH_BLOCKING=1 (which raised the fuzz's failure rate to 4 of 12)
cd $GRAFT_REPO_ROOT
B=tests/stress_handles.bin
run() { echo "== $*"; ( timeout 300 "$@" 2>&1; echo "rc=$?" ) | grep -v amdgpu.ids | tail -${TAILN:-8}; }
IT=${IT:-40}
export HIP_LAUNCH_BLOCKING=1
run $B --handles 3 --threads 3 --iters $IT --work elem10 --kind 2 --mode spawn --quiet
run $B --handles 3 --threads 3 --iters $IT --work elem10 --kind 2 --mode spawn --fresh 1 --quiet
run $B --handles 3 --threads 3 --iters $IT --work elem10 --kind 2 --mode lockstep --quiet
run $B --handles 4 --threads 4 --iters $IT --work mix --mode lockstep --quiet
run $B --handles 4 --threads 4 --iters $IT --work mix --mode threads --fresh 1 --quiet
run $B --handles 4 --threads 4 --iters $IT --work fd --mode lockstep --quiet
run $B --handles 4 --threads 4 --iters $IT --work trip --mode lockstep --quiet
run $B --handles 4 --threads 4 --iters $IT --work fem4 --mode lockstep --quiet
