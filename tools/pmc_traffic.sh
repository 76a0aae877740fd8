#!/bin/bash
# HBM traffic per kernel (rocprofv3 PMC, one counter per pass, two eager episodes of workload W = S (default) or C):
# writes gpurun_out/pmc/${TAG}_pmc_{train,eval}_{FETCH_SIZE,WRITE_SIZE}.txt and ${TAG}_pmc_traffic_${W}.json (copy the
# ones to be judged into profiles/).  Run on the GPU box from the repository root:  bash tools/pmc_traffic.sh [S|C] [TAG]
set -e
W=${1:-S}
TAG=${2:-r02}
export R3D_WORKLOAD=$W R3D_TAG=$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
for mode in train eval; do
  for c in FETCH_SIZE WRITE_SIZE; do
    dir="/tmp/pmc_${W}_${mode}_$c"
    rm -rf "$dir"
    if ! timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "r3d_" -d "$dir" -o r -- python3 tools/one_episode.py $mode $W > "$dir.log" 2>&1; then
      echo "rocprofv3 failed ($mode $c):"; tail -20 "$dir.log"; exit 1
    fi
    db=$(ls "$dir"/*.db 2>/dev/null | tail -1)
    if [ -z "$db" ]; then echo "no rocprofv3 database under $dir"; tail -20 "$dir.log"; exit 1; fi
    python3 tools/pmc_summary.py "$db" $c gpurun_out/pmc/${W}_${mode}_$c.json > gpurun_out/pmc/${TAG}_pmc_${W}_${mode}_$c.txt
  done
done
python3 - <<'P'
import json, os, re
W, TAG = os.environ["R3D_WORKLOAD"], os.environ["R3D_TAG"]
out = {"workload": W, "_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace --kernel-include-regex r3d_ -- python3 tools/one_episode.py {train,eval} WORKLOAD (tools/pmc_traffic.sh); KB per launch, raw counters (gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x, other widths uncalibrated: MI355X_MICROARCH.md)"}
for mode in ("train", "eval"):
    f = json.load(open("gpurun_out/pmc/%s_%s_FETCH_SIZE.json" % (W, mode))); w = json.load(open("gpurun_out/pmc/%s_%s_WRITE_SIZE.json" % (W, mode)))
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
json.dump(out, open("gpurun_out/pmc/%s_pmc_traffic_%s.json" % (TAG, W), "w"), indent=1, sort_keys=True)
print({k: v for k, v in out["train"].items() if k.startswith("r3d_cg_")})
P
