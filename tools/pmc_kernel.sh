#!/bin/bash
# SQ counters of one kernel (regex) while a python tool runs:  bash tools/pmc_kernel.sh REGEX tools/x.py [args]
# One rocprofv3 pass per counter pair (kernel-trace only beside --pmc); prints per-dispatch averages.
REGEX=$1
shift
cd /tmp && export TMPDIR=/tmp
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf /tmp/pk/$n
  (cd $GRAFT_REPO_ROOT && timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$REGEX" -d /tmp/pk/$n -o r -- python3 "$@" > /tmp/pk_$n.log 2>&1) || { echo FAIL $c; tail -3 /tmp/pk_$n.log; continue; }
  python3 - "$(ls /tmp/pk/$n/*.db | tail -1)" <<'P'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
for n, k, v, d in c.execute("select counter_name, count(*), sum(counter_value), avg(duration) from pmc_events group by counter_name"):
    print("%-28s dispatches %4d  per-dispatch %16.0f  avg_dur_ns %.0f" % (n, k, v / max(k, 1), d))
P
done
