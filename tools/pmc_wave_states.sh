#!/bin/bash
# Where do the waves of one kernel spend their cycles?  SQ wave-state counters (MI355X_MICROARCH.md, "rocprofv3 PMC slots":
# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES) for the kernels matching a regex, one command.
# usage (GPU box): bash tools/pmc_wave_states.sh OUT_TAG KERNEL_REGEX -- command...
TAG=$1; RX=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/wpmc
  if ! timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "$RX" -d /tmp/wpmc -o r -- "$@" > /tmp/wpmc.log 2>&1; then echo "failed: $grp"; tail -3 /tmp/wpmc.log; continue; fi
  db=$(ls /tmp/wpmc/*.db | tail -1)
  for c in $grp; do python3 tools/pmc_summary.py "$db" $c gpurun_out/pmc/${TAG}_$c.json | head -3 | tail -2 | cut -c1-60,73-110; done
done
