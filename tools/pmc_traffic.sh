#!/bin/bash
# HBM traffic per kernel (rocprofv3 PMC, one counter per pass -- FETCH_SIZE and WRITE_SIZE do not fit one pass --, three
# episode-batched steps of workload W = S (default) or C): writes gpurun_out/pmc/${TAG}_pmc_${W}_{train,eval}_{FETCH_SIZE,
# WRITE_SIZE}.txt and ${TAG}_pmc_traffic_${W}.json (copy the ones to be judged into profiles/).  The JSON carries, per
# kernel, raw KB per launch and the calibration of the counters on known byte counts (tools/one_step.py's two 1 GiB
# copies: 4 B and 16 B per lane).  Run on the GPU box from the repository root:  bash tools/pmc_traffic.sh [S|C] [TAG] [modes]
set -e
W=${1:-S}
TAG=${2:-r04}
shift 2 || true
MODES=${@:-train}
export R3D_WORKLOAD=$W R3D_TAG=$TAG R3D_MODES="$MODES"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
E=32; [ "$W" = "C" ] && E=8
for mode in $MODES; do
  for c in FETCH_SIZE WRITE_SIZE; do
    dir="/tmp/pmc_${W}_${mode}_$c"
    rm -rf "$dir"
    if ! timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -d "$dir" -o r -- python3 tools/one_step.py $mode $W $E 3 > "$dir.log" 2>&1; then
      echo "rocprofv3 failed ($mode $c):"; tail -20 "$dir.log"; exit 1
    fi
    db=$(ls "$dir"/*.db 2>/dev/null | tail -1)
    if [ -z "$db" ]; then echo "no rocprofv3 database under $dir"; tail -20 "$dir.log"; exit 1; fi
    python3 tools/pmc_summary.py "$db" $c gpurun_out/pmc/${W}_${mode}_$c.json > gpurun_out/pmc/${TAG}_pmc_${W}_${mode}_$c.txt
  done
done
python3 - <<'P'
import json, os, re
W, TAG, MODES = os.environ["R3D_WORKLOAD"], os.environ["R3D_TAG"], os.environ["R3D_MODES"].split()
GIB_KB = float(1 << 20)
out = {"workload": W, "_provenance": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/one_step.py MODE WORKLOAD E 3 (tools/pmc_traffic.sh); KB per launch, RAW counters.  calibration: counter reading / true KB of two 1 GiB streaming copies (4 B per lane: r3d_copy_cols_kernel; 16 B per lane: torch's vectorised copy) -- MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of wide (16 B per lane) coalesced reads on gfx950, other widths are to be calibrated on a known byte count"}
for mode in MODES:
    f = json.load(open("gpurun_out/pmc/%s_%s_FETCH_SIZE.json" % (W, mode))); w = json.load(open("gpurun_out/pmc/%s_%s_WRITE_SIZE.json" % (W, mode)))
    d = {}
    calib = {}
    for name, v in f.items():
        if name.startswith("r3d_copy_cols_kernel") and v["max"] > 0.2 * GIB_KB:  # the 1 GiB calibration copy is its largest launch
            calib["4B_per_lane_fetch_reading_over_true"] = round(v["max"] / GIB_KB, 4)
            if name in w: calib["4B_per_lane_write_reading_over_true"] = round(w[name]["max"] / GIB_KB, 4)
        short = re.sub(r"^void ", "", name).split("(")[0].split("<")[0]
        if not short.startswith("r3d_"):
            if "vectorized_elementwise_kernel" in name and v["total"] / max(v["calls"], 1) > 0.2 * GIB_KB:
                calib["16B_per_lane_fetch_reading_over_true"] = round(v["avg"] / GIB_KB, 4)
                if name in w: calib["16B_per_lane_write_reading_over_true"] = round(w[name]["avg"] / GIB_KB, 4)
            continue
        e = d.setdefault(short, {"fetch_kb_per_launch": 0.0, "write_kb_per_launch": 0.0, "launches": 0, "_f": 0.0, "_w": 0.0})
        e["_f"] += v["total"]; e["launches"] += v["calls"]
        if name in w: e["_w"] += w[name]["total"]
    for short, e in d.items():
        n = max(e["launches"], 1)
        e["fetch_kb_per_launch"] = round(e.pop("_f") / n, 1); e["write_kb_per_launch"] = round(e.pop("_w") / n, 1)
    out[mode] = dict(sorted(d.items()))
    out[mode + "_calibration"] = calib
json.dump(out, open("gpurun_out/pmc/%s_pmc_traffic_%s.json" % (TAG, W), "w"), indent=1, sort_keys=True)
for mode in MODES:
    print(mode, "calibration", out[mode + "_calibration"])
    print({k: v for k, v in out[mode].items() if k.startswith("r3d_cg_") or k.startswith("r3d_edgeconv_bwd1") or k.startswith("r3d_pointwise")})
P
