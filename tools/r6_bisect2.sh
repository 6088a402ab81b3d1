#!/bin/bash
# round 6: bisect 2 -- parts 0 (4-byte keys, direct colptr) + 1 (packed keys, radix tier, colptr by scan) side by side corrupt part 1
cd $GRAFT_REPO_ROOT
B=tests/stress_handles.bin
N=${N:-6}
trial() {  # tag, env, args...
  tag=$1; envs=$2; shift 2
  ok=0; bad=0; why=""
  for i in $(seq 1 $N); do
    if env $envs timeout 120 $B --work parts --kind 2 --iters 15 --quiet "$@" > gpurun_out/bis.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); why="$why | $(grep -v 'amdgpu.ids\|Broken pipe\|oredump\|core dump' gpurun_out/bis.log | head -1 | cut -c1-100)"; fi
  done
  echo "$tag: ok $ok fail $bad $why"
}
trial alone1_hog_lds A=1 --handles 1 --parts 1 --mode serial --hog lds
trial alone1_hog_fill A=1 --handles 1 --parts 1 --mode serial --hog fill
trial alone0_hog_lds A=1 --handles 1 --parts 0 --mode serial --hog lds
trial spawn_01 A=1 --handles 2 --threads 2 --mode spawn --parts 0,1
trial spawn_01_f13_on0 A=1 --handles 2 --threads 2 --mode spawn --parts 0,1 --force0 13
trial spawn_01_f13_on1 A=1 --handles 2 --threads 2 --mode spawn --parts 0,1 --force1 13
trial spawn_01_f3_on1 A=1 --handles 2 --threads 2 --mode spawn --parts 0,1 --force1 3
trial spawn_01_f2_on1 A=1 --handles 2 --threads 2 --mode spawn --parts 0,1 --force1 2
trial spawn_01_f2_on0 A=1 --handles 2 --threads 2 --mode spawn --parts 0,1 --force0 2
trial spawn_01_f14_on0 A=1 --handles 2 --threads 2 --mode spawn --parts 0,1 --force0 14
trial spawn_01_onequeue GPU_MAX_HW_QUEUES=1 --handles 2 --threads 2 --mode spawn --parts 0,1
trial spawn_01_serialize AMD_SERIALIZE_KERNEL=3 --handles 2 --threads 2 --mode spawn --parts 0,1
trial spawn_10 A=1 --handles 2 --threads 2 --mode spawn --parts 1,0
trial spawn_21 A=1 --handles 2 --threads 2 --mode spawn --parts 2,1
