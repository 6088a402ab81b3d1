#!/bin/bash
# round 6: a long concurrent-handles campaign on the final tree (tests/stress_handles.bin): every workload, every mode, with and
# without allocation churn, beside the LDS / bandwidth hogs, under blocking launches
cd $GRAFT_REPO_ROOT
B=tests/stress_handles.bin
ok=0; bad=0
run() { if ( timeout 600 "$@" --quiet > gpurun_out/sf.log 2>&1 ); then ok=$((ok+1)); else bad=$((bad+1)); echo "FAIL: $* :: $(grep -v 'amdgpu.ids\|Broken pipe\|oredump\|core dump' gpurun_out/sf.log | head -2 | cut -c1-200)"; fi; }
IT=${IT:-60}
for m in threads lockstep spawn; do
  for w in mix parts sum elem10 fem4 trip fd; do
    run $B --handles 6 --threads 6 --iters $IT --work $w --mode $m --kind 2
  done
done
run $B --handles 8 --threads 8 --iters $IT --work mix --mode threads --fresh 1
run $B --handles 8 --threads 8 --iters $IT --work parts --mode lockstep --fresh 1 --kind 2
run $B --handles 6 --threads 6 --iters $IT --work parts --mode spawn --hog lds --kind 2
run $B --handles 6 --threads 6 --iters $IT --work mix --mode lockstep --hog fill
HIP_LAUNCH_BLOCKING=1 run $B --handles 6 --threads 6 --iters $IT --work parts --mode spawn --kind 2
HIP_LAUNCH_BLOCKING=1 run $B --handles 6 --threads 6 --iters $IT --work mix --mode lockstep
AMD_SERIALIZE_KERNEL=3 run $B --handles 6 --threads 6 --iters $IT --work parts --mode lockstep --kind 2

echo "stress campaign: ok $ok fail $bad"
