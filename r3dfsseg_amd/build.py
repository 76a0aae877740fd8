"""Builds the gfx950 shared library r3dfsseg_amd/libr3d_hip.so with hipcc.

    python -m r3dfsseg_amd.build

hipcc cross-compiles without a GPU.  -ffp-contract=off: the bit-exact index kernels
spell out every fused multiply-add (``__builtin_fmaf``); nothing else may fuse.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libr3d_hip.so")
SOURCES = ["error.hip", "knn.hip", "gemm.hip", "gemm_bx3.hip", "edgeconv.hip", "attention.hip", "head_proto.hip",
           "head_graph.hip", "aux_heads.hip", "train_ops.hip", "edgeconv_train.hip", "contrast.hip"]
# No packed fp32 vector arithmetic (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32), neither from the SLP vectoriser nor from
# float2 / float4 source arithmetic: measured on MI355X (profiles/r02_experiments.md, section 9), a wave executing them
# beside waves of a bf16-MFMA-dense kernel on the same SIMD got wrong results in lanes 48-63 of one half of the pair --
# the label-propagation graph weights came out wrong in ~20 % of the solves that ran beside the bf16 x 3 attention
# kernels, and never once in a build without these instructions.  (The host pass prints "'-packed-fp32-ops' is not a
# recognized feature" for the second flag: it only applies to the device pass.)
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdr = [os.path.join(CSRC, h) for h in ("common.h", "edge_tile.h", "edgeconv_bwd_bx3.h")]
    objs = []
    procs = []
    for s in [x for x in SOURCES if os.path.exists(os.path.join(CSRC, x))]:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdr):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (s, out))
        elif out.strip() and verbose:
            print(out)
    if failed:
        raise RuntimeError("r3dfsseg_amd: HIP build failed")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
