#!/bin/bash
# Wave-state survey of the batched training step: for every kernel the share of its wave cycles spent waiting on memory /
# barriers (WAIT_ANY), waiting to issue (WAIT_INST_ANY) and issuing (ACTIVE_INST_ANY), from one rocprofv3 PMC pass
# (MI355X_MICROARCH.md, "rocprofv3 PMC slots").  usage (GPU box, repository root): bash tools/pmc_wave_survey.sh [TAG]
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
rm -rf /tmp/wsurvey
if ! timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/wsurvey -o r -- python3 tools/one_step.py train S 32 2 > /tmp/wsurvey.log 2>&1; then
  echo "rocprofv3 failed"; tail -20 /tmp/wsurvey.log; exit 1
fi
db=$(ls /tmp/wsurvey/*.db | tail -1)
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY; do python3 tools/pmc_summary.py "$db" $c /tmp/ws_$c.json > /dev/null; done
python3 - "$TAG" <<'P'
import json, sys
tag = sys.argv[1]
d = {c: json.load(open("/tmp/ws_%s.json" % c)) for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}
rows = []
for k, v in d["SQ_WAVE_CYCLES"].items():
    w = v["total"]
    if w <= 0 or not ("r3d_" in k):
        continue
    g = lambda c: d[c].get(k, {"total": 0})["total"] / w
    rows.append((w, k, v["calls"], g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY"), g("SQ_ACTIVE_INST_ANY")))
rows.sort(reverse=True)
out = ["# rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python3 tools/one_step.py train S 32 2 (tools/pmc_wave_survey.sh)",
       "%-72s %6s %14s %9s %10s %8s" % ("kernel", "calls", "wave cycles", "WAIT_ANY", "WAIT_INST", "ACTIVE")]
for w, k, n, a, b, c in rows[:40]:
    out.append("%-72s %6d %14.0f %8.1f%% %9.1f%% %7.1f%%" % (k[:72], n, w, 100 * a, 100 * b, 100 * c))
open("gpurun_out/pmc/%s_wave_survey.txt" % tag, "w").write("\n".join(out) + "\n")
print("\n".join(out))
P
