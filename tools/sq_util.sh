cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/sq2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline ${1:---no-extra} > gpurun_out/sq2.log 2>&1
python3 tools/sq_counters.py gpurun_out/sq2 | head -${2:-4} | cut -c1-500
rm -rf gpurun_out/sq2
