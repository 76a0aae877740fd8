#!/bin/bash
# The judged evidence of a build in one GPU call: kernel statistics (train, eval) and PMC traffic of the batched step,
# copied into profiles/ ON THE BOX so that the bench lines that follow quote them, then the bench lines themselves
# (train S with the CPU baseline, eval S, train C).  Everything lands under gpurun_out/ (copy into profiles/ afterwards).
#   bash tools/final_evidence.sh [TAG] [a|b]      a: profiles + PMC + train bench;  b: eval and workload-C bench lines
set -e
TAG=${1:-r04}
PART=${2:-a}
cd "$GRAFT_REPO_ROOT"
if [ "$PART" = "b" ]; then
  python bench.py --mode eval > gpurun_out/${TAG}_bench_eval_S.json 2> gpurun_out/bench_eval_S.log
  echo "bench eval S done"
  python bench.py --workload C --no-cpu-baseline > gpurun_out/${TAG}_bench_train_C.json 2> gpurun_out/bench_train_C.log
  python tools/bench_summary.py gpurun_out/${TAG}_bench_eval_S.json gpurun_out/${TAG}_bench_train_C.json
  exit 0
fi
mkdir -p gpurun_out/prof gpurun_out/pmc
bash tools/profile.sh $TAG S train eval > gpurun_out/prof_final.log 2>&1
cp gpurun_out/prof/${TAG}_rocprofv3_kernel_stats_train_batched_S.txt gpurun_out/prof/${TAG}_rocprofv3_kernel_stats_eval_batched_S.txt profiles/
echo "kernel statistics done" 
bash tools/pmc_traffic.sh S $TAG train > gpurun_out/pmc_final.log 2>&1
cp gpurun_out/pmc/${TAG}_pmc_traffic_S.json gpurun_out/pmc/${TAG}_pmc_S_train_FETCH_SIZE.txt gpurun_out/pmc/${TAG}_pmc_S_train_WRITE_SIZE.txt profiles/
echo "PMC passes done"
python bench.py > gpurun_out/${TAG}_bench_train_S.json 2> gpurun_out/bench_train_S.log
echo "bench train S done"
python tools/bench_summary.py gpurun_out/${TAG}_bench_train_S.json
