"""Research script (not product, not test): PCG iteration counts of candidate preconditioners for the
reduced (Schur) pose system on synthetic graphs, on the CPU with scipy.  Used to choose what to build
as HIP kernels next."""
import sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl
sys.path.insert(0, ".")
from oracle import oracle
from tests import util
from toyslam_amd import synth


def assemble(g):
    o = util.to_oracle(g)
    e, A, B = oracle.edge_eval(o)
    P = g.n_poses; L = g.n_landmarks
    # poses are ids 0..P-1, landmarks P.. in synth graphs
    et = o.e_type; ids = o.e_ids.astype(np.int64); w = o.e_inf
    m_lm = et == 1
    chi = (e * e * w).sum(1)
    hw = np.where(chi <= 2.25, 1.0, 1.5 / np.sqrt(np.maximum(chi, 1e-300)))
    # LM edges
    El = int(m_lm.sum())
    Al = A[m_lm, :6].reshape(El, 2, 3); Bl = B[m_lm, :4].reshape(El, 2, 2); el = e[m_lm, :2]
    Wl = (w[m_lm, :2] * hw[m_lm, None])
    i_p = ids[m_lm, 0]; i_l = ids[m_lm, 1] - P
    AtW = np.einsum('ekr,ek->erk', Al, Wl)            # (E,3,2) = A^T W
    Hpp_e = np.einsum('erk,ekc->erc', AtW, Al)        # 3x3
    Hpl_e = np.einsum('erk,ekc->erc', AtW, Bl)        # 3x2
    BtW = np.einsum('ekr,ek->erk', Bl, Wl)
    Hll_e = np.einsum('erk,ekc->erc', BtW, Bl)        # 2x2
    bp_e = -np.einsum('erk,ek->er', AtW, el); bl_e = -np.einsum('erk,ek->er', BtW, el)
    def blocks_to_coo(bi, bj, blk, nr, nc, shape):
        E = len(bi)
        r = (bi[:, None, None] * nr + np.arange(nr)[None, :, None]) + np.zeros((1, 1, nc), int)
        c = (bj[:, None, None] * nc + np.arange(nc)[None, None, :]) + np.zeros((1, nr, 1), int)
        return sp.coo_matrix((blk.ravel(), (r.ravel(), c.ravel())), shape=shape).tocsr()
    Hpp = blocks_to_coo(i_p, i_p, Hpp_e, 3, 3, (3 * P, 3 * P))
    Hpl = blocks_to_coo(i_p, i_l, Hpl_e, 3, 2, (3 * P, 2 * L))
    Hll = blocks_to_coo(i_l, i_l, Hll_e, 2, 2, (2 * L, 2 * L))
    bp = np.zeros(3 * P); np.add.at(bp.reshape(P, 3), i_p, bp_e)
    bl = np.zeros(2 * L); np.add.at(bl.reshape(L, 2), i_l, bl_e)
    # ODOM
    m_od = ~m_lm
    Wo = w[m_od] * hw[m_od, None]; eo = e[m_od]; a = ids[m_od, 0]; b = ids[m_od, 1]
    D = np.zeros((len(a), 3, 3)); D[:, [0, 1, 2], [0, 1, 2]] = Wo
    Hpp = Hpp + blocks_to_coo(a, a, D, 3, 3, (3 * P, 3 * P)) + blocks_to_coo(b, b, D, 3, 3, (3 * P, 3 * P)) \
        - blocks_to_coo(a, b, D, 3, 3, (3 * P, 3 * P)) - blocks_to_coo(b, a, D, 3, 3, (3 * P, 3 * P))
    np.add.at(bp.reshape(P, 3), a, Wo * eo); np.add.at(bp.reshape(P, 3), b, -Wo * eo)
    g0 = sp.coo_matrix(([1e6] * 3, ([0, 1, 2], [0, 1, 2])), shape=(3 * P, 3 * P)).tocsr()
    Hpp = Hpp + g0
    # Schur
    Hll = Hll.tocsc()
    Hll_inv = spl.inv(Hll) if L < 3000 else None
    # block-diagonal inverse of Hll (2x2 blocks)
    d = Hll.tobsr((2, 2)); dd = np.zeros((L, 2, 2))
    d.sort_indices()
    for i in range(L):
        for k in range(d.indptr[i], d.indptr[i + 1]):
            if d.indices[k] == i: dd[i] = d.data[k]
    inv = np.linalg.inv(dd)
    Hll_inv = sp.bsr_matrix((inv, np.arange(L), np.arange(L + 1)), shape=(2 * L, 2 * L)).tocsr()
    S = (Hpp - Hpl @ Hll_inv @ Hpl.T).tocsr()
    rhs = bp - Hpl @ (Hll_inv @ bl)
    return S, rhs


def pcg(S, b, M, tol=1e-10, maxit=20000):
    it = [0]
    def cb(xk): it[0] += 1
    t = time.time()
    x, info = spl.cg(S, b, rtol=tol, atol=0, maxiter=maxit, M=M, callback=cb)
    return it[0], time.time() - t, np.linalg.norm(S @ x - b) / np.linalg.norm(b)


def block_jacobi(S, bs):
    n = S.shape[0]; nb = (n + bs - 1) // bs
    lus = []
    Sc = S.tocsc()
    for k in range(nb):
        i0, i1 = k * bs, min(n, (k + 1) * bs)
        lus.append(np.linalg.inv(Sc[i0:i1, i0:i1].toarray()))
    def mv(r):
        z = np.empty_like(r)
        for k in range(nb):
            i0, i1 = k * bs, min(n, (k + 1) * bs); z[i0:i1] = lus[k] @ r[i0:i1]
        return z
    return spl.LinearOperator(S.shape, matvec=mv)


def banded(S, wblocks):
    C = S.tocoo(); keep = np.abs(C.row // 3 - C.col // 3) <= wblocks
    Bm = sp.csc_matrix((C.data[keep], (C.row[keep], C.col[keep])), shape=S.shape)
    lu = spl.splu(Bm)
    return spl.LinearOperator(S.shape, matvec=lu.solve), Bm


def rigid_Z(P, xy, m):
    """aggregation prolongator: segments of m poses, 3 rigid modes each"""
    nagg = (P + m - 1) // m
    rows, cols, vals = [], [], []
    for a in range(nagg):
        i0, i1 = a * m, min(P, (a + 1) * m)
        c = xy[i0:i1].mean(0)
        for i in range(i0, i1):
            rows += [3 * i, 3 * i + 1, 3 * i, 3 * i + 1, 3 * i + 2]
            cols += [3 * a, 3 * a + 1, 3 * a + 2, 3 * a + 2, 3 * a + 2]
            vals += [1.0, 1.0, -(xy[i, 1] - c[1]), (xy[i, 0] - c[0]), 1.0]
    return sp.csr_matrix((vals, (rows, cols)), shape=(3 * P, 3 * nagg))


def two_level(S, Z, smoother, nu=1):
    Sc = (Z.T @ S @ Z).tocsc(); lu = spl.splu(Sc)
    def mv(r):
        # symmetric multiplicative: pre-smooth, coarse, post-smooth
        z = smoother.matvec(r)
        res = r - S @ z
        z = z + Z @ lu.solve(Z.T @ res)
        res = r - S @ z
        z = z + smoother.matvec(res)
        return z
    return spl.LinearOperator(S.shape, matvec=mv), Sc


def multilevel(S, xy, ms, omega=1.0):
    """V(1,1) cycle, unsmoothed rigid aggregation with factors ms per level, block-Jacobi(3) smoother"""
    levels = []
    A = S; pts = xy
    for m in ms:
        P = A.shape[0] // 3
        Z = rigid_Z(P, pts, m)
        levels.append((A, block_jacobi(A, 3), Z))
        A = (Z.T @ A @ Z).tocsr()
        nagg = A.shape[0] // 3
        pts = np.array([pts[a * m:min(P, (a + 1) * m)].mean(0) for a in range(nagg)])
    lu = spl.splu(A.tocsc())
    def cycle(l, r):
        if l == len(levels): return lu.solve(r)
        A, sm, Z = levels[l]
        z = omega * sm.matvec(r)
        z = z + Z @ cycle(l + 1, Z.T @ (r - A @ z))
        z = z + omega * sm.matvec(r - A @ z)
        return z
    return spl.LinearOperator(S.shape, matvec=lambda r: cycle(0, r)), A.shape[0]


if __name__ == "__main__" and len(sys.argv) <= 2:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    g = synth.make(n, 10)
    t = time.time(); S, b = assemble(g); print("assembled", S.shape, "nnz", S.nnz, time.time() - t)
    xy = g.v_pos[:g.n_poses, :2]
    print("block-jacobi 3:", pcg(S, b, block_jacobi(S, 3)))
    for m in (16, 64):
        print("segment block-jacobi m=%d:" % m, pcg(S, b, block_jacobi(S, 3 * m)))
    for wb in (8, 16):
        M, Bm = banded(S, wb)
        print("banded w=%d (nnz kept %.2f):" % (wb, Bm.nnz / S.nnz), pcg(S, b, M))
    for m in (8, 32):
        Z = rigid_Z(g.n_poses, xy, m)
        M, Sc = two_level(S, Z, block_jacobi(S, 3))
        print("two-level m=%d (coarse n=%d nnz=%d) BJ3 smoother:" % (m, Sc.shape[0], Sc.nnz), pcg(S, b, M))
    for ms in ((4, 4, 4, 4), (8, 8, 8), (2, 2, 2, 2, 2, 2, 2)):
        M, nc = multilevel(S, xy, ms)
        print("multilevel", ms, "coarsest n=%d:" % nc, pcg(S, b, M))


def multilevel2(S, xy, ms, nu=1, gamma=1, smooth_P=False, omega=0.7, sm_block=3):
    """gamma=1 V-cycle, gamma=2 W-cycle; optional prolongator smoothing (smoothed aggregation)."""
    levels = []
    A = S; pts = xy
    for m in ms:
        P = A.shape[0] // 3
        Z = rigid_Z(P, pts, m)
        if smooth_P:
            D = A.tobsr((3, 3)); D.sort_indices()
            dinv = np.zeros((P, 3, 3))
            for i in range(P):
                for k in range(D.indptr[i], D.indptr[i + 1]):
                    if D.indices[k] == i: dinv[i] = np.linalg.inv(D.data[k])
            Dinv = sp.bsr_matrix((dinv, np.arange(P), np.arange(P + 1)), shape=A.shape).tocsr()
            Z = (Z - omega * (Dinv @ (A @ Z))).tocsr()
        levels.append((A, block_jacobi(A, sm_block), Z))
        A = (Z.T @ A @ Z).tocsr()
        nagg = A.shape[0] // 3
        pts = np.array([pts[a * m:min(P, (a + 1) * m)].mean(0) for a in range(nagg)])
    lu = spl.splu(A.tocsc())
    nnz = [l[0].nnz for l in levels] + [A.nnz]
    def cycle(l, r):
        if l == len(levels): return lu.solve(r)
        A, sm, Z = levels[l]
        z = np.zeros_like(r)
        for _ in range(nu): z = z + sm.matvec(r - A @ z)
        for g_ in range(gamma if l > 0 or True else 1):
            z = z + Z @ cycle(l + 1, Z.T @ (r - A @ z))
        for _ in range(nu): z = z + sm.matvec(r - A @ z)
        return z
    return spl.LinearOperator(S.shape, matvec=lambda r: cycle(0, r)), nnz


def study2(n):
    g = synth.make(n, 10)
    S, b = assemble(g); xy = g.v_pos[:g.n_poses, :2]
    print("n poses", n)
    for name, kw in [("V(1,1) 8,8,8", dict(ms=(8, 8, 8))), ("V(2,2) 8,8,8", dict(ms=(8, 8, 8), nu=2)),
                     ("W(1,1) 8,8,8", dict(ms=(8, 8, 8), gamma=2)), ("W(1,1) 4,4,4,4,4", dict(ms=(4, 4, 4, 4, 4), gamma=2)),
                     ("SA V(1,1) 8,8,8", dict(ms=(8, 8, 8), smooth_P=True)),
                     ("SA V(1,1) 4,4,4,4", dict(ms=(4, 4, 4, 4), smooth_P=True)),
                     ("W(1,1) 8,8,8,8", dict(ms=(8, 8, 8, 8), gamma=2))]:
        M, nnz = multilevel2(S, xy, **kw)
        print(name, "nnz per level", nnz, pcg(S, b, M, tol=1e-8))

if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "x":
    study2(int(sys.argv[1]))


def multilevel3(S, xy, ms, smooth_levels, omega=0.7):
    """like multilevel2 V(1,1) but prolongator smoothing only on the levels listed in smooth_levels"""
    levels = []
    A = S; pts = xy
    for li, m in enumerate(ms):
        P = A.shape[0] // 3
        Z = rigid_Z(P, pts, m)
        if li in smooth_levels:
            D = A.tobsr((3, 3)); D.sort_indices()
            dinv = np.zeros((P, 3, 3))
            for i in range(P):
                for k in range(D.indptr[i], D.indptr[i + 1]):
                    if D.indices[k] == i: dinv[i] = np.linalg.inv(D.data[k])
            Dinv = sp.bsr_matrix((dinv, np.arange(P), np.arange(P + 1)), shape=A.shape).tocsr()
            Z = (Z - omega * (Dinv @ (A @ Z))).tocsr()
        levels.append((A, block_jacobi(A, 3), Z))
        A = (Z.T @ A @ Z).tocsr()
        nagg = A.shape[0] // 3
        pts = np.array([pts[a * m:min(P, (a + 1) * m)].mean(0) for a in range(nagg)])
    lu = spl.splu(A.tocsc())
    nnz = [l[0].nnz for l in levels] + [A.nnz]
    def cycle(l, r):
        if l == len(levels): return lu.solve(r)
        A, sm, Z = levels[l]
        z = sm.matvec(r)
        z = z + Z @ cycle(l + 1, Z.T @ (r - A @ z))
        z = z + sm.matvec(r - A @ z)
        return z
    return spl.LinearOperator(S.shape, matvec=lambda r: cycle(0, r)), nnz


def study3(n):
    g = synth.make(n, 10)
    S, b = assemble(g); xy = g.v_pos[:g.n_poses, :2]
    for name, ms, sl in [("unsmoothed L0 (m=4), SA above 8,8,8", (4, 8, 8, 8), {1, 2, 3}),
                         ("unsmoothed L0 (m=8), SA above 8,8", (8, 8, 8), {1, 2}),
                         ("unsmoothed L0 (m=2), SA above 8,8,8", (2, 8, 8, 8), {1, 2, 3}),
                         ("SA all 8,8,8", (8, 8, 8), {0, 1, 2}),
                         ("SA all 16,8,8", (16, 8, 8), {0, 1, 2}),
                         ("SA all 8,8,8 omega .5", (8, 8, 8), {0, 1, 2})]:
        M, nnz = multilevel3(S, xy, ms, sl, omega=0.5 if "omega" in name else 0.7)
        print(name, nnz, pcg(S, b, M, tol=1e-8), flush=True)

if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "3":
    study3(int(sys.argv[1]))
