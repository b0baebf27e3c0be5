"""oracle/so3_numpy.py -- TEST INFRASTRUCTURE ONLY (checker; never imported by gpr_calculator_amd).

numpy/scipy restatement of the reference's SO(3) power-spectrum descriptor
(gpr_calc/SO3.py:186-323 `calculate`, :348-407 neighbour bookkeeping, :409-453 cutoff / radial basis /
quadrature, :521-605 Bessel values, :608-727 `compute_dcs`).

Pinned by tests/golden/so3_*.npz, which tests/golden/make_golden_python.py produced by running the
reference's own SO3.calculate (all numerics reference code; only the ASE neighbour search, a
third-party dependency absent here, was supplied).  The neighbour search itself (ASE >= 3.23
NeighborList, bothways, no self interaction, skin 0) is restated below as a brute-force image
loop: "parity unpinned" for that third-party piece, checked for consistency against the
product's independent implementation in tests/test_neighbors.py.
"""
import numpy as np
from scipy.special import spherical_in

try:                                   # scipy >= 1.15
    from scipy.special import sph_harm_y

    def _ylm(m, l, phi, theta):
        return sph_harm_y(l, m, theta, phi)
except ImportError:                    # pragma: no cover
    from scipy.special import sph_harm as _ylm


def neighbor_pairs_bruteforce(positions, cell, pbc, rcut):
    """(centre, neighbour, r_ij) for every |r_ij| < rcut; periodic self images kept, i==j zero image dropped."""
    pos = np.asarray(positions, float)
    cell = np.asarray(cell, float).reshape(3, 3)
    pbc = np.asarray(pbc, bool)
    n = len(pos)
    nimg = [0, 0, 0]
    if pbc.any():
        vol = abs(np.linalg.det(cell))
        for k in range(3):
            if pbc[k]:
                h = vol / np.linalg.norm(np.cross(cell[(k + 1) % 3], cell[(k + 2) % 3]))
                nimg[k] = int(np.ceil(rcut / h)) + 1          # +1: positions may sit outside the cell
    cs, js, vs = [], [], []
    for i in range(n):
        for j in range(n):
            for a in range(-nimg[0], nimg[0] + 1):
                for b in range(-nimg[1], nimg[1] + 1):
                    for c in range(-nimg[2], nimg[2] + 1):
                        if i == j and a == 0 and b == 0 and c == 0:
                            continue
                        v = pos[j] + a * cell[0] + b * cell[1] + c * cell[2] - pos[i]
                        if v @ v < rcut * rcut:
                            cs.append(i); js.append(j); vs.append(v)
    return np.array(cs, int), np.array(js, int), np.array(vs, float).reshape(-1, 3)


def cosine_cutoff(r, rc, derivative=False):                       # SO3.py:409-415
    if not derivative:
        return 0.5 * (np.cos(np.pi * r / rc) + 1.0)
    return -0.5 * np.pi / rc * np.sin(np.pi * r / rc)


def W(nmax):                                                      # SO3.py:417-430
    arr = np.zeros((nmax, nmax))
    for a in range(1, nmax + 1):
        t1 = (2 * a + 5) * (2 * a + 6) * (2 * a + 7)
        for b in range(1, a + 1):
            t2 = (2 * b + 5) * (2 * b + 6) * (2 * b + 7)
            arr[a - 1, b - 1] = arr[b - 1, a - 1] = np.sqrt(t1 * t2) / (5 + a + b) / (6 + a + b) / (7 + a + b)
    sinv = np.linalg.inv(arr)
    eigvals, V = np.linalg.eig(sinv)
    return np.dot(np.dot(V, np.diag(np.sqrt(eigvals))), np.linalg.inv(V))


def radial_table(nmax, lmax, rcut, alpha):
    """quadrature nodes r_q [NQ] and G[n, q] = g_n(r_q) * w_q * r_q^2 exp(-alpha r_q^2) sqrt(1-t_q^2)
    (SO3.py:446-453, 432-444, 623-655)"""
    NQ = (nmax + lmax + 1) * 10
    t = np.cos((2 * np.arange(1, NQ + 1) - 1) * np.pi / 2 / NQ)
    weight = np.pi / NQ * rcut / 2
    rq = rcut / 2 * (t + 1)
    w = W(nmax)
    G = np.zeros((nmax, NQ))
    for n in range(1, nmax + 1):
        for a in range(1, nmax + 1):
            phi = (rcut - rq) ** (a + 2) / np.sqrt(2 * rcut ** (2 * a + 7) / (2 * a + 5) / (2 * a + 6) / (2 * a + 7))
            G[n - 1] += w[n - 1, a - 1] * phi
        G[n - 1] *= weight
    G *= rq ** 2 * np.exp(-alpha * rq ** 2) * np.sqrt(1 - t ** 2)
    return rq, G


def compute_dcs(vec, nmax, lmax, rcut, alpha):
    """expansion coefficients C[P, n, l, m] (complex) and their gradients dC[P, n, l, m, 3]
    w.r.t. the neighbour position (SO3.py:608-727)."""
    P = len(vec)
    r = np.linalg.norm(vec, axis=1)
    u = vec / r[:, None]
    rq, G = radial_table(nmax, lmax, rcut, alpha)
    arg = 2 * alpha * np.outer(r, rq)                              # [P, NQ]
    ls = np.arange(lmax + 1)
    bes = np.stack([spherical_in(l, arg) for l in ls], axis=2)     # [P, NQ, L]
    dbes = np.stack([spherical_in(l, arg, derivative=True) for l in ls], axis=2)
    integ = np.einsum("nq,pql->pnl", G, bes)                       # [P, n, L]
    dinteg_r = np.einsum("nq,pql->pnl", G * (2 * alpha * rq)[None, :], dbes)   # d/dr of the radial integral
    ginteg = dinteg_r[..., None] * u[:, None, None, :]             # [P, n, L, 3]
    expo = 4 * np.pi * np.exp(-alpha * r ** 2)
    gexpo = (-2 * alpha * r * expo)[:, None] * u
    fc = cosine_cutoff(r, rcut)
    gfc = cosine_cutoff(r, rcut, True)[:, None] * u
    theta = np.arccos(vec[:, 2] / r)
    phi = np.arctan2(vec[:, 1], vec[:, 0])
    Ms = 2 * (lmax + 1) + 1
    ms = 2 * lmax + 1
    Y = np.zeros((P, lmax + 2, Ms), complex)
    for l in range(lmax + 2):
        for m in range(-l, l + 1):
            Y[:, l, Ms // 2 + m] = _ylm(m, l, phi, theta)
    gY = np.zeros((P, lmax + 1, ms, 3), complex)
    for l in range(1, lmax + 1):
        for m in range(-l, l + 1):
            mi, Mi = ms // 2 + m, Ms // 2 + m
            x0 = -np.sqrt(((l + 1) ** 2 - m ** 2) / (2 * l + 1) / (2 * l + 3)) * l * Y[:, l + 1, Mi] / r
            if abs(m) <= l - 1:
                x0 += np.sqrt((l ** 2 - m ** 2) / (2 * l - 1) / (2 * l + 1)) * (l + 1) * Y[:, l - 1, Mi] / r
            xp = -np.sqrt((l + m + 1) * (l + m + 2) / 2 / (2 * l + 1) / (2 * l + 3)) * l * Y[:, l + 1, Mi + 1] / r
            if abs(m + 1) <= l - 1:
                xp -= np.sqrt((l - m - 1) * (l - m) / 2 / (2 * l - 1) / (2 * l + 1)) * (l + 1) * Y[:, l - 1, Mi + 1] / r
            xm = -np.sqrt((l - m + 1) * (l - m + 2) / 2 / (2 * l + 1) / (2 * l + 3)) * l * Y[:, l + 1, Mi - 1] / r
            if abs(m - 1) <= l - 1:
                xm -= np.sqrt((l + m - 1) * (l + m) / 2 / (2 * l - 1) / (2 * l + 1)) * (l + 1) * Y[:, l - 1, Mi - 1] / r
            gY[:, l, mi, 0] = 1 / np.sqrt(2) * (xm - xp)
            gY[:, l, mi, 1] = 1j / np.sqrt(2) * (xm + xp)
            gY[:, l, mi, 2] = x0
    Y = Y[:, :lmax + 1, 1:1 + ms]
    Yin = np.einsum("plm,pnl->pnlm", Y, integ)
    C = expo[:, None, None, None] * Yin
    dC = gexpo[:, None, None, None, :] * Yin[..., None]
    dC = dC + expo[:, None, None, None, None] * (np.einsum("plmk,pnl->pnlmk", gY, integ)
                                                 + np.einsum("plm,pnlk->pnlmk", Y, ginteg))
    dC = dC * fc[:, None, None, None, None] + gfc[:, None, None, None, :] * C[..., None]
    C = C * fc[:, None, None, None]
    return C, dC


def so3_calculate(numbers, positions, cell, pbc, nmax=3, lmax=4, rcut=5.0, alpha=2.0, stress=False):
    """{'x': [n, ncoef], 'dxdr': [S, ncoef, 3], 'seq': [S, 2], 'rdxdr': None | [S, ncoef, 3, 3]}
    following SO3.calculate (SO3.py:186-323) with weight_on=False."""
    numbers = np.asarray(numbers)
    positions = np.asarray(positions, float)
    n = len(numbers)
    ci, cj, vec = neighbor_pairs_bruteforce(positions, cell, pbc, rcut)
    ncoef = nmax * (nmax + 1) // 2 * (lmax + 1)
    tril = np.tril_indices(nmax)
    ls = np.arange(lmax + 1)
    norm = np.sqrt(2 * np.sqrt(2) * np.pi / np.sqrt(2 * ls + 1))
    # seq: per centre, sorted unique neighbour ids with the centre itself inserted (SO3.py:386-399)
    seq = []
    for i in range(n):
        ids = set(cj[ci == i].tolist())
        ids.add(i)
        seq.extend([i, j] for j in sorted(ids))
    seq = np.array(seq, dtype=np.int64).reshape(-1, 2)
    x = np.zeros((n, ncoef))
    dxdr = np.zeros((len(seq), ncoef, 3))
    pstress = np.zeros((len(seq), ncoef, 3, 3))
    if len(ci):
        C, dC = compute_dcs(vec, nmax, lmax, rcut, alpha)
        wgt = numbers[cj].astype(float)                              # SO3.py:385 (weight_on False)
        C = C * wgt[:, None, None, None] * norm[None, None, :, None]
        dC = dC * wgt[:, None, None, None, None] * norm[None, None, :, None, None]
    row_of = {(int(a), int(b)): k for k, (a, b) in enumerate(seq)}
    for i in range(n):
        sel = np.nonzero(ci == i)[0]
        if len(sel) == 0:
            continue
        ctot = C[sel].sum(axis=0)                                    # [n, l, m]
        Pm = np.einsum("ijk,ljk->ilj", ctot, np.conj(ctot)).real
        x[i] = Pm[tril].flatten()
        dP = np.einsum("wijkn,ljk->wiljn", dC[sel], np.conj(ctot))
        dP = (dP + np.conj(np.transpose(dP, (0, 2, 1, 3, 4)))).real   # [w, n1, n2, l, 3]
        Rj = vec[sel] + positions[i]
        for w_, p in enumerate(sel):
            k = row_of[(i, int(cj[p]))]
            contrib = dP[w_][tril].reshape(ncoef, 3)
            dxdr[k] += contrib
            pstress[k] -= np.einsum("a,cb->cab", Rj[w_], contrib)
        rows_i = [row_of[(i, int(j))] for j in sorted(set(cj[sel].tolist()) | {i})]
        kii = row_of[(i, i)]
        dxdr[kii] -= dxdr[rows_i].sum(axis=0)
        tot = dP.sum(axis=0)[tril].reshape(ncoef, 3)
        pstress[kii] += np.einsum("a,cb->cab", positions[i], tot)
    out = {"x": x, "dxdr": dxdr, "seq": seq, "rdxdr": None}
    if stress:
        out["rdxdr"] = -pstress / abs(np.linalg.det(np.asarray(cell, float).reshape(3, 3)))
    return out
