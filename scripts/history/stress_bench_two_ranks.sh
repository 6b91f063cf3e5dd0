#!/bin/bash
# repeats the driver's N = 2 launch on one GPU; keeps the stderr of any failing run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stress2
fails=0
for i in $(seq 1 ${N:-12}); do
  port=$((29600 + i))
  t0=$(date +%s)
  BENCH_ONE_GPU=1 BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu > gpurun_out/stress2/out_$i.json 2> gpurun_out/stress2/err_$i.txt
  rc=$?
  t1=$(date +%s)
  echo "run $i rc=$rc $((t1 - t0)) s lines=$(wc -l < gpurun_out/stress2/out_$i.json)"
  if [ $rc -eq 0 ]; then rm -f gpurun_out/stress2/err_$i.txt; else fails=$((fails + 1)); fi
done
echo "failures: $fails"
