#!/bin/bash
# rocprofv3 kernel trace of any python tool of this repository:  bash tools/prof_tool.sh TAG tools/ab_attention.py [args]
# -> gpurun_out/prof/${TAG}_kernel_stats.txt
set -e
TAG=$1
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
dir="/tmp/prof_${TAG}"
rm -rf "$dir"
if ! timeout -k 10 400 rocprofv3 --kernel-trace -d "$dir" -o r -- python3 "$@" > "$dir.log" 2>&1; then
  echo "rocprofv3 failed:"; tail -20 "$dir.log"; exit 1
fi
db=$(ls "$dir"/*.db 2>/dev/null | tail -1)
out=gpurun_out/prof/${TAG}_kernel_stats.txt
echo "# rocprofv3 --kernel-trace -- python3 $@" > $out
python3 tools/prof_summary.py "$db" 40 >> $out
tail -8 "$dir.log"
head -24 $out | cut -c1-150
