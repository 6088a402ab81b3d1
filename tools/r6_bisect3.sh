#!/bin/bash
# round 6: bisect 3 (experiments build): is it the runtime's memset, or the read-back copy, that loses its place in the stream?
export ESP_EXTRA_FLAGS=-DESP_EXPERIMENTS
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
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
trial base A=1 --handles 2 --threads 2 --mode spawn --parts 0,1
trial ownfill ESP_X_OWNFILL=1 --handles 2 --threads 2 --mode spawn --parts 0,1
trial sync_after_memset ESP_X_SYNC_AFTER_MEMSET=1 --handles 2 --threads 2 --mode spawn --parts 0,1
trial sync_before_readback ESP_X_SYNC_BEFORE_READBACK=1 --handles 2 --threads 2 --mode spawn --parts 0,1
trial base_again A=1 --handles 2 --threads 2 --mode spawn --parts 0,1
