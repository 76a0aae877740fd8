#!/bin/bash
# HBM traffic per kernel (rocprofv3 PMC, one counter per pass, two eager episodes of workload S): writes
# profiles/r01_pmc_{train,eval}_{FETCH_SIZE,WRITE_SIZE}.txt and profiles/r01_pmc_traffic.json.  Run on the GPU box from
# the repository root:  bash tools/pmc_traffic.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
for mode in train eval; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$mode_$c
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "r3d_" -d /tmp/pmc_${mode}_$c -o r -- python3 tools/one_episode.py $mode > /tmp/pmc_${mode}_$c.log 2>&1
    python3 tools/pmc_summary.py $(ls /tmp/pmc_${mode}_$c/*.db | tail -1) $c gpurun_out/pmc/${mode}_$c.json > gpurun_out/pmc/r01_pmc_${mode}_$c.txt
  done
done
python3 - <<'P'
import json, re
out = {"_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace --kernel-include-regex r3d_ -- python3 tools/one_episode.py {train,eval} (tools/pmc_traffic.sh); KB per launch, raw counters (gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x, other widths uncalibrated: MI355X_MICROARCH.md)"}
for mode in ("train", "eval"):
    f = json.load(open("gpurun_out/pmc/%s_FETCH_SIZE.json" % mode)); w = json.load(open("gpurun_out/pmc/%s_WRITE_SIZE.json" % mode))
    d = {}
    for name, v in f.items():
        short = re.sub(r"^void ", "", name).split("(")[0].split("<")[0]
        e = d.setdefault(short, {"fetch_kb_per_launch": 0.0, "write_kb_per_launch": 0.0, "launches_in_2_episodes": 0, "_f": 0.0, "_w": 0.0})
        e["_f"] += v["total"]; e["launches_in_2_episodes"] += v["calls"]
        if name in w: e["_w"] += w[name]["total"]
    for short, e in d.items():
        n = max(e["launches_in_2_episodes"], 1)
        e["fetch_kb_per_launch"] = round(e.pop("_f") / n, 1); e["write_kb_per_launch"] = round(e.pop("_w") / n, 1)
    out[mode] = dict(sorted(d.items()))
json.dump(out, open("gpurun_out/pmc/r01_pmc_traffic.json", "w"), indent=1, sort_keys=True)
print({k: out["train"][k] for k in ("r3d_cg_spmv_kernel", "r3d_cg_update_kernel")})
P
