#!/bin/bash
# builds and runs the ablation probe for each bit combination given (default: a standard set)
set -e
cd "$GRAFT_REPO_ROOT"
for bits in ${@:-0 128}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DGB_ABL=$bits -I r3dfsseg_amd/csrc tools/probe/gemm_bx3_abl.hip -o /tmp/gb_abl_$bits 2>/dev/null
  timeout -k 5 60 /tmp/gb_abl_$bits
done
