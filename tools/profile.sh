#!/bin/bash
# rocprofv3 kernel trace of the eager single-episode schedule of bench.py (one episode in flight: kernel times do not
# overlap) -> gpurun_out/prof/${TAG}_rocprofv3_kernel_stats_${mode}_eager_${W}.txt (copy the ones to be judged into
# profiles/).  Run on the GPU box from the repository root:  bash tools/profile.sh TAG [S|C] [train|eval ...]
set -e
TAG=${1:-r02}
W=${2:-S}
shift 2 || true
MODES=${@:-train eval}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
for mode in $MODES; do
  dir="/tmp/prof_${W}_${mode}"
  rm -rf "$dir"
  steps=20; [ "$W" = "C" ] && steps=6
  if ! timeout -k 10 400 rocprofv3 --kernel-trace -d "$dir" -o r -- python3 bench.py --mode $mode --workload $W --slots 0 \
        --steps $steps --warmup 3 --no-cpu-baseline --steady-steps 0 > "$dir.log" 2>&1; then
    echo "rocprofv3 failed ($mode):"; tail -20 "$dir.log"; exit 1
  fi
  db=$(ls "$dir"/*.db 2>/dev/null | tail -1)
  if [ -z "$db" ]; then echo "no rocprofv3 database under $dir"; tail -20 "$dir.log"; exit 1; fi
  out=gpurun_out/prof/${TAG}_rocprofv3_kernel_stats_${mode}_eager_${W}.txt
  echo "# rocprofv3 --kernel-trace -- python3 bench.py --mode $mode --workload $W --slots 0 --steps $steps --warmup 3 --no-cpu-baseline --steady-steps 0 (tools/profile.sh, tools/prof_summary.py)" > $out
  python3 tools/prof_summary.py "$db" 60 >> $out
  grep '^{' "$dir.log" | tail -1 > gpurun_out/prof/${TAG}_bench_${mode}_eager_${W}.json || true
  head -12 $out
done
