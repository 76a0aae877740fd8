#!/bin/bash
# rocprofv3 kernel trace of the HEADLINE schedule (episode-batched steps: one launch sequence per 32 episodes, so kernel
# times do not overlap) -> gpurun_out/prof/${TAG}_rocprofv3_kernel_stats_${mode}_batched_${W}.txt (copy the ones to be
# judged into profiles/).  Run on the GPU box from the repository root:  bash tools/profile.sh TAG [S|C] [train|eval ...]
set -e
TAG=${1:-r04}
W=${2:-S}
shift 2 || true
MODES=${@:-train eval}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
for mode in $MODES; do
  dir="/tmp/prof_${W}_${mode}"
  rm -rf "$dir"
  E=32; [ "$W" = "C" ] && E=8
  if ! timeout -k 10 400 rocprofv3 --kernel-trace -d "$dir" -o r -- python3 tools/one_step.py $mode $W $E 5 > "$dir.log" 2>&1; then
    echo "rocprofv3 failed ($mode):"; tail -20 "$dir.log"; exit 1
  fi
  db=$(ls "$dir"/*.db 2>/dev/null | tail -1)
  if [ -z "$db" ]; then echo "no rocprofv3 database under $dir"; tail -20 "$dir.log"; exit 1; fi
  out=gpurun_out/prof/${TAG}_rocprofv3_kernel_stats_${mode}_batched_${W}.txt
  echo "# rocprofv3 --kernel-trace -- python3 tools/one_step.py $mode $W $E 5   (5 steps of $E episodes in ONE launch sequence each + 2 calibration copies; tools/profile.sh, tools/prof_summary.py)" > $out
  python3 tools/prof_summary.py "$db" 70 >> $out
  head -14 $out
done
