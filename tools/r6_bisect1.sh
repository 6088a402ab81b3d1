#!/bin/bash
# round 6: bisecting the reproduced corruption (three uneven parts of a ten-node mesh flushed side by side): which partners, which
# host pattern, which kernels
cd $GRAFT_REPO_ROOT
B=tests/stress_handles.bin
N=${N:-6}
trial() {  # tag, args...
  tag=$1; shift
  ok=0; bad=0; why=""
  for i in $(seq 1 $N); do
    if timeout 120 $B --work parts --kind 2 --iters 15 --quiet "$@" > gpurun_out/bis.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); why="$why | $(grep -v 'amdgpu.ids\|Broken pipe\|oredump\|core dump' gpurun_out/bis.log | head -1 | cut -c1-110)"; fi
  done
  echo "$tag: ok $ok fail $bad $why"
}
trial spawn_012 --handles 3 --threads 3 --mode spawn
trial spawn_111 --handles 3 --threads 3 --mode spawn --parts 1,1,1
trial spawn_11 --handles 2 --threads 2 --mode spawn --parts 1,1
trial spawn_01 --handles 2 --threads 2 --mode spawn --parts 0,1
trial spawn_02 --handles 2 --threads 2 --mode spawn --parts 0,2
trial spawn_000 --handles 3 --threads 3 --mode spawn --parts 0,0,0
trial serial_012 --handles 3 --mode serial
trial interleave_012 --handles 3 --mode interleave
trial threads_012 --handles 3 --threads 3 --mode threads
for f in 2 3 13 14 15 24 26 30; do
  trial spawn_012_force$f --handles 3 --threads 3 --mode spawn --force $f
done
