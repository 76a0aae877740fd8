#!/bin/bash
# which partial products the gradient bars of the full-size parity test need: the test under each R3D_GEMM_BX3 mask
cd "$GRAFT_REPO_ROOT"
for mask in ${@:-0 1 2 3}; do
  echo "=== R3D_GEMM_BX3=$mask"
  R3D_GEMM_BX3=$mask timeout -k 10 280 python -m pytest tests/test_gpu_parity_full.py -x -q -s -k config2_full_size_training 2>&1 | grep -A9 "full-size encoder gradients against" | grep -v "^\s*print\|^\s*for\|np.median" | head -12
  R3D_GEMM_BX3=$mask timeout -k 10 280 python -m pytest tests/test_gpu_parity_full.py -x -q -k config2_full_size_training 2>&1 | tail -n 1
done
