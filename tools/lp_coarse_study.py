"""Offline study on dumped label-propagation systems (tools/dump_lp_systems.py): iterations of the library's two-level CG
(A-DEF2, coarse space W = D^1/2 [indicators of M aggregates], aggregates = nearest of M evenly subsampled prototypes in
feature space; csrc/head_graph.hip section 5) as a function of M and of the seed choice, in fp32 like the device.
usage: lp_coarse_study.py FILE.npz [...]"""
import sys
import numpy as np
import scipy.sparse as sp

ALPHA, TOL = 0.99, 1e-6


def aggregates(nodes, n_proto, M, how="even"):
    P = nodes[:n_proto].astype(np.float32)
    if how == "even":
        src = (np.arange(M, dtype=np.int64) * (n_proto - 1)) // (M - 1)
    elif how == "fps":  # farthest point sampling over the prototypes
        src = [0]
        d = ((P - P[0]) ** 2).sum(1)
        for _ in range(M - 1):
            j = int(d.argmax()); src.append(j)
            d = np.minimum(d, ((P - P[j]) ** 2).sum(1))
        src = np.array(src)
    seeds = P[src]
    x = nodes.astype(np.float32)
    d2 = (x ** 2).sum(1)[:, None] - 2 * x @ seeds.T + (seeds ** 2).sum(1)[None]
    return d2.argmin(1)


def solve(S, dinv, Y, agg, M, maxit=400):
    n = S.shape[0]
    f = np.float32
    u = (1.0 / dinv).astype(f)
    W = sp.csr_matrix((u, (np.arange(n), agg)), shape=(n, M), dtype=f)
    A = (sp.identity(n, dtype=f, format="csr") - f(ALPHA) * S).astype(f)
    MW = (A @ W).toarray().astype(f)
    Wd = W.toarray()
    E = (Wd.T.astype(np.float64) @ MW.astype(np.float64))
    live = np.abs(E).sum(0) > 0  # empty aggregates
    Einv = np.zeros_like(E)
    Einv[np.ix_(live, live)] = np.linalg.inv(E[np.ix_(live, live)])
    Einv = Einv.astype(f)
    b = Y.astype(f)
    x = Wd @ (Einv @ (Wd.T @ b))
    r = b - A @ x

    def prec(r):
        return r + Wd @ (Einv @ (Wd.T @ r - MW.T @ r))
    z = prec(r)
    p = z.copy()
    rz = (r * z).sum(0)
    bb = (b * b).sum(0)
    for it in range(1, maxit + 1):
        q = A @ p
        pq = (p * q).sum(0)
        a = np.where(pq != 0, rz / np.where(pq != 0, pq, 1), 0).astype(f)
        x += a * p
        r -= a * q
        rr = (r * r).sum(0)
        if np.all(rr <= TOL * TOL * bb):
            return it, x
        z = prec(r)
        rz_new = (r * z).sum(0)
        beta = np.where(rz != 0, rz_new / np.where(rz != 0, rz, 1), 0).astype(f)
        p = z + beta * p
        rz = rz_new
    return maxit, x


for path in sys.argv[1:]:
    g = np.load(path)
    n = int(g["n"]); n_proto = int(g["n_proto"])
    S = sp.csr_matrix((g["val"], g["col"], g["row_ptr"]), shape=(n, n), dtype=np.float32)
    nodes = g["nodes"]; Y = g["Y"]; dinv = g["dinv"]
    print("%s: n %d, prototypes %d, nnz %d, device iterations %s" % (path, n, n_proto, S.nnz, g["stats"]))
    for how in ("even", "fps"):
        for M in (64, 96, 128, 192, 256, n_proto):
            if M > n_proto:
                continue
            it, x = solve(S, dinv, Y, aggregates(nodes, n_proto, M, how), M)
            err = np.abs(x - g["Z"]).max() / np.abs(g["Z"]).max()
            print("   seeds %-4s M %3d: %3d iterations   (solution against the device's: %.1e)" % (how, M, it, err))
