cd /tmp && export TMPDIR=/tmp
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "r3d_pointwise_gemm" -d /tmp/pg/$n -o r -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py 20480 512 256 > /tmp/pg_$n.log 2>&1 || { echo FAIL $c; tail -3 /tmp/pg_$n.log; continue; }
  python3 - "$(ls /tmp/pg/$n/*.db | tail -1)" <<'P'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
for n, k, v, d in c.execute("select counter_name, count(*), sum(counter_value), avg(duration) from pmc_events group by counter_name"):
    print("%-28s rows %5d  sum %16.0f  per-dispatch %14.0f  avg_dur_ns %.0f" % (n, k, v, v / 20, d))
P
done
