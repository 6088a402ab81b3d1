cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for st in 1 2 3 0; do
  rm -rf gpurun_out/sqs
  ESP_LOCAL_STOP=$st rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/sqs -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/sqs.log 2>&1
  echo "stop=$st"; python tools/sq_counters.py gpurun_out/sqs | grep local_k | cut -c1-300
done
rm -rf gpurun_out/sqs
