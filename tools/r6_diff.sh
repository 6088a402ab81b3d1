#!/bin/bash
# round 6: WHERE the corrupted result departs from the handle's result alone
cd $GRAFT_REPO_ROOT
B=tests/stress_handles.bin
for i in $(seq 1 ${N:-10}); do
  timeout 120 $B --work parts --kind 2 --iters 6 --quiet --handles 2 --threads 2 --mode spawn --parts 0,1 2>&1 | grep -v 'amdgpu.ids\|Broken pipe\|oredump\|core dump' | cut -c1-1500 | head -4
  echo "--"
done
