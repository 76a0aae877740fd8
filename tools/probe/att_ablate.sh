#!/bin/bash
# Timing ablations of r3d_attention_fwd_bx3_kernel: builds libr3d_abl_<mask>.so with -DATT_ABL=<mask> (run HERE, hipcc
# cross-compiles) -> tools/probe/; `run` on the GPU box times each with tools/ab_attention.py.
# bits: 1 no LDS-DMA, 2 plain instead of transposed LDS reads, 4 no MFMA (results are wrong with any bit set)
set -e
cd "$(dirname "$0")/../.."
MASKS=${MASKS:-"0 1 2 4"}
if [ "$1" = "build" ]; then
  for m in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -DATT_ABL=$m -c r3dfsseg_amd/csrc/attention.hip -o /tmp/att_abl_$m.o &
  done
  wait
  for m in $MASKS; do
    objs=$(ls r3dfsseg_amd/csrc/*.o | grep -v attention.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libr3d_abl_$m.so $objs /tmp/att_abl_$m.o
  done
else
  [ "$1" = "run" ] && shift
  for m in $MASKS; do
    echo "== ATT_ABL=$m"
    R3D_LIB=$PWD/tools/probe/libr3d_abl_$m.so python3 "$@" 2>&1 | tail -1
  done
fi
