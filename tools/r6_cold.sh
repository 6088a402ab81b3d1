#!/bin/bash
# round 6: is it the FIRST launch of a kernel, made by several host threads at once?  Cold processes: no serial warm-up, the
# concurrent phase comes first (references from a warm run's file)
cd $GRAFT_REPO_ROOT
B=tests/stress_handles.bin
N=${N:-30}
trial() {  # tag, env, args...
  tag=$1; envs=$2; shift 2
  $B "$@" --iters 1 --quiet --writeref gpurun_out/ref_$tag.txt > /dev/null 2>&1
  ok=0; bad=0
  for i in $(seq 1 $N); do
    if env $envs timeout 120 $B "$@" --iters 2 --quiet --cold 1 --ref gpurun_out/ref_$tag.txt > gpurun_out/cold.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); echo "$tag: FAIL $(grep -v amdgpu.ids gpurun_out/cold.log | head -3 | cut -c1-220)"; fi
  done
  echo "$tag ($envs): ok $ok fail $bad"
}
trial elem10_spawn A=1 --handles 3 --threads 3 --work elem10 --kind 2 --mode spawn
trial elem10_spawn_blocking HIP_LAUNCH_BLOCKING=1 --handles 3 --threads 3 --work elem10 --kind 2 --mode spawn
trial mix_lockstep_blocking HIP_LAUNCH_BLOCKING=1 --handles 4 --threads 4 --work mix --mode lockstep
trial fd_lockstep_blocking HIP_LAUNCH_BLOCKING=1 --handles 4 --threads 4 --work fd --mode lockstep
