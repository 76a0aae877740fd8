cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt 2>/dev/null
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  n=$(echo $c | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "r3d_knn_append" -d /tmp/pk/$n -o r -- $GRAFT_REPO_ROOT/tools/knnbench/kb 64 20 0 > /tmp/pk_$n.log 2>&1 || { echo FAIL $c; tail -3 /tmp/pk_$n.log; continue; }
  python3 - "$(ls /tmp/pk/$n/*.db | tail -1)" <<'P'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
for n, k, v, d in c.execute("select counter_name, count(*), sum(counter_value), avg(duration) from pmc_events group by counter_name"):
    print("%-28s rows %5d  sum %16.0f  avg_dur_ns %.0f" % (n, k, v, d))
P
done
