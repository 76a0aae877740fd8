"""Generates tests/golden/episode_ref.h5 with the REFERENCE's own `write_episode` under real h5py (build container only).

    /opt/conda/bin/python3.9 oracle/gen_golden_h5.py        # the interpreter of the image that has h5py (3.3.0, HDF5 1.10.6)

TEST INFRASTRUCTURE.  `dataloaders/loader.py` cannot be imported by that interpreter (no torch, open3d, transforms3d
there), and the interpreter that has torch has no h5py.  `write_episode` / `read_episode` (loader.py:1687-1721) use
nothing but h5py and print, so their two function definitions are taken out of the file with `ast` and compiled on
their own with `h5 = h5py`: the code that writes the fixture is the reference's, unmodified.  The episode is
deterministic (numpy RandomState below; the test regenerates it).  `--check FILE` reads FILE with the reference's
read_episode and prints a digest (tests/test_episode_h5.py hands it files written by r3dfsseg_amd/h5lite.py).
"""
import ast
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/dataloaders/loader.py"


def reference_functions():
    import h5py
    tree = ast.parse(open(REF).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("write_episode", "read_episode")]
    assert len(keep) == 2
    ns = {"h5": h5py, "np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return ns["write_episode"], ns["read_episode"]


def episode(seed=21, n_way=2, k_shot=3, N=96):
    """The 8-tuple of loader.py:1688 (deliberately in the dtypes a sampler hands over: float64 clouds, int64 masks --
    write_episode's dtype= arguments do the conversion)."""
    rs = np.random.RandomState(seed)
    return (rs.uniform(0, 1, (n_way, k_shot, N, 9)), rs.randint(0, 2, (n_way, k_shot, N)),
            rs.uniform(0, 1, (n_way, N, 9)), rs.randint(0, n_way + 1, (n_way, N)), np.array([3, 8]),
            rs.randint(0, 5, (n_way, k_shot, N)), rs.randint(0, 5, (n_way, N)), rs.randint(0, 2, (n_way, k_shot, N)))


def digest(arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(("%s %s " % (a.dtype.str, a.shape)).encode())
        h.update(a.tobytes())
    return h.hexdigest()


if __name__ == "__main__":
    write_episode, read_episode = reference_functions()
    if len(sys.argv) == 3 and sys.argv[1] == "--check":
        print(digest(read_episode(sys.argv[2])))
    else:
        out = os.path.join(ROOT, "tests", "golden", "episode_ref.h5")
        write_episode(out, episode())
        print(out, os.path.getsize(out), digest(read_episode(out)))
