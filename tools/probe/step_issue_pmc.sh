#!/bin/bash
# Instruction mix per kernel of the batched training step: VALU / MFMA / LDS / VMEM / scalar instructions issued and the
# matrix pipe's busy cycles -> which kernels are bound by vector ISSUE (tools/probe/step_issue_summary.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  rm -rf /tmp/spmc
  if ! timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/spmc -o r -- python3 tools/one_step.py train S 32 2 > /tmp/spmc.log 2>&1; then echo "failed: $grp"; tail -5 /tmp/spmc.log; continue; fi
  db=$(ls /tmp/spmc/*.db | tail -1)
  for c in $grp; do python3 tools/pmc_summary.py "$db" $c gpurun_out/pmc/issue_$c.json > /dev/null; done
done
python3 - <<'P'
import json
g = lambda c: json.load(open("gpurun_out/pmc/issue_%s.json" % c))
V, Mf, L, VM, SA, SM, MB, B = (g(c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"))
rows = []
for k in V:
    m = Mf.get(k, {}).get("total", 0)
    if m <= 0: continue
    tot = lambda d: d.get(k, {}).get("total", 0.0)
    rows.append((tot(MB), k, tot(V) / m, tot(L) / m, tot(VM) / m, (tot(SA) + tot(SM)) / m, tot(MB) / m))
print("%-60s %8s %8s %8s %8s %10s" % ("kernel", "VALU/MFMA", "LDS", "VMEM", "scalar", "cyc/MFMA"))
for r in sorted(rows, reverse=True)[:16]:
    print("%-60s %8.2f %8.2f %8.2f %8.2f %10.1f" % (r[1][:60], r[2], r[3], r[4], r[5], r[6]))
P
