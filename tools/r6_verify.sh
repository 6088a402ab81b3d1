#!/bin/bash
# round 6: the reproducer (parts 0 + 1 of a ten-node mesh flushed side by side) on the current build
cd $GRAFT_REPO_ROOT
B=tests/stress_handles.bin
N=${N:-8}
trial() {  # tag, env, args...
  tag=$1; envs=$2; shift 2
  ok=0; bad=0; why=""
  for i in $(seq 1 $N); do
    if env $envs timeout 120 $B --work parts --kind 2 --iters ${IT:-15} --quiet "$@" > gpurun_out/bis.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); why="$why | $(grep -v 'amdgpu.ids\|Broken pipe\|oredump\|core dump' gpurun_out/bis.log | head -1 | cut -c1-160)"; fi
  done
  echo "$tag: ok $ok fail $bad $why"
}
trial spawn_01 A=1 --handles 2 --threads 2 --mode spawn --parts 0,1
trial spawn_012 A=1 --handles 3 --threads 3 --mode spawn
trial spawn_21 A=1 --handles 2 --threads 2 --mode spawn --parts 2,1
trial spawn_012_blocking HIP_LAUNCH_BLOCKING=1 --handles 3 --threads 3 --mode spawn
trial lockstep_012012 A=1 --handles 6 --threads 6 --mode lockstep
