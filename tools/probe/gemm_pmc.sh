#!/bin/bash
# SQ counters of the point-wise GEMM kernels (tools/gemm_bench.py), one rocprofv3 --pmc pass per group
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/gpmc
  if ! timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/gpmc -o r -- python3 tools/gemm_bench.py > /tmp/gpmc.log 2>&1; then echo "failed: $grp"; tail -5 /tmp/gpmc.log; continue; fi
  db=$(ls /tmp/gpmc/*.db | tail -1)
  for c in $grp; do
    python3 tools/pmc_summary.py "$db" $c /tmp/x.json | grep -E "kernel|bx3p_kernel<4>|pointwise_gemm_kernel\(" | cut -c1-120
  done
done
