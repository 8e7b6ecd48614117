"""Deterministic synthetic SPD / Hermitian-PD test problems (BASELINE.json configs 2-5,
definitions from SURVEY.md section 8(d)).

Every generator returns the LOWER triangle in CSC form with int64 indices
(`stype = -1` in CHOLMOD terms, the form cholmod_super_numeric consumes,
CHOLMOD/Supernodal/cholmod_super_numeric.c:127-134), sorted row indices, plus the
grid shape so that a geometric nested-dissection ordering can be built.
"""
import numpy as np


def _csc_lower_from_coo(n, rows, cols, vals):
    """Build sorted lower-triangular CSC from COO (entries with row >= col only)."""
    keep = rows >= cols
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    order = np.lexsort((rows, cols))
    rows, cols, vals = rows[order], cols[order], vals[order]
    # sum duplicates
    if rows.size:
        key = cols.astype(np.int64) * n + rows
        uniq, start = np.unique(key, return_index=True)
        if uniq.size != key.size:
            vals = np.add.reduceat(vals, start)
            rows, cols = rows[start], cols[start]
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.add.at(Ap, cols + 1, 1)
    Ap = np.cumsum(Ap)
    return Ap, rows.astype(np.int64), vals


def grid_index(shape):
    """Linear index of grid node; x fastest."""
    return np.arange(int(np.prod(shape)), dtype=np.int64).reshape(shape[::-1])


def laplacian(shape, diag=None, off=-1.0, dtype=np.float64):
    """d-dimensional (2D 5-point / 3D 7-point) Laplacian on a regular grid.

    shape = (nx, ny) or (nx, ny, nz); diagonal 2*d (4 in 2D, 6 in 3D) unless given.
    """
    d = len(shape)
    n = int(np.prod(shape))
    if diag is None:
        diag = 2.0 * d
    idx = grid_index(shape)                      # axes: (z, y, x)
    rows = [np.arange(n, dtype=np.int64)]
    cols = [np.arange(n, dtype=np.int64)]
    vals = [np.full(n, diag, dtype=dtype)]
    for ax in range(d):
        lo = [slice(None)] * d
        hi = [slice(None)] * d
        lo[ax] = slice(0, -1)
        hi[ax] = slice(1, None)
        a = idx[tuple(lo)].ravel()
        b = idx[tuple(hi)].ravel()               # b > a
        rows.append(b)
        cols.append(a)
        vals.append(np.full(a.size, off, dtype=dtype))
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    vals = np.concatenate(vals)
    Ap, Ai, Ax = _csc_lower_from_coo(n, rows, cols, vals)
    return dict(n=n, Ap=Ap, Ai=Ai, Ax=Ax, shape=tuple(shape), dof=1, stype=-1)


def laplacian_2d(k):
    return laplacian((k, k))


def laplacian_3d(k):
    return laplacian((k, k, k))


def helmholtz_3d(k, diag=6.5, phase=0.3):
    """Complex Hermitian 3D 7-point: real diagonal `diag`, lower off-diagonals -exp(i*phase)."""
    return laplacian((k, k, k), diag=diag, off=-np.exp(1j * phase), dtype=np.complex128)


def elasticity27(k, dof=3):
    """3D 27-point stencil, `dof` unknowns per node.

    node-diagonal dof x dof block: 33 on the diagonal, -0.5 off; every one of the <= 26
    neighbour blocks: -1 on its diagonal, -0.1 off.  Strictly diagonally dominant => SPD.
    """
    shape = (k, k, k)
    nn = k ** 3
    n = nn * dof
    idx = grid_index(shape)
    rows, cols, vals = [], [], []
    a = np.arange(nn, dtype=np.int64)
    for p in range(dof):
        for q in range(dof):
            rows.append(a * dof + p)
            cols.append(a * dof + q)
            vals.append(np.full(nn, 33.0 if p == q else -0.5))
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if (dz, dy, dx) <= (0, 0, 0):
                    continue                      # take each unordered pair once (b > a)
                sl_a, sl_b = [], []
                for dd in (dz, dy, dx):
                    if dd == 1:
                        sl_a.append(slice(0, -1)); sl_b.append(slice(1, None))
                    elif dd == -1:
                        sl_a.append(slice(1, None)); sl_b.append(slice(0, -1))
                    else:
                        sl_a.append(slice(None)); sl_b.append(slice(None))
                na = idx[tuple(sl_a)].ravel()
                nb = idx[tuple(sl_b)].ravel()
                for p in range(dof):
                    for q in range(dof):
                        v = -1.0 if p == q else -0.1
                        r = nb * dof + p
                        c = na * dof + q
                        # keep lower triangle: swap if needed (matrix is symmetric)
                        lo = np.minimum(r, c)
                        hi = np.maximum(r, c)
                        rows.append(hi); cols.append(lo)
                        vals.append(np.full(na.size, v))
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    vals = np.concatenate(vals)
    # off-diagonal node blocks appear once per (p,q); the symmetric partner (q,p) of the
    # transposed block maps to the same lower entry only when p != q AND nodes swap, so
    # no duplicates arise except inside the node-diagonal block, where (p,q),(q,p) with
    # p != q both land on (max,min): halve by keeping row >= col of the original.
    Ap, Ai, Ax = _csc_lower_from_coo_nodup(n, rows, cols, vals)
    return dict(n=n, Ap=Ap, Ai=Ai, Ax=Ax, shape=shape, dof=dof, stype=-1)


def _csc_lower_from_coo_nodup(n, rows, cols, vals):
    """As _csc_lower_from_coo but duplicates are collapsed (first wins), not summed."""
    keep = rows >= cols
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    key = cols.astype(np.int64) * n + rows
    uniq, first = np.unique(key, return_index=True)
    rows, cols, vals = rows[first], cols[first], vals[first]
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.add.at(Ap, cols + 1, 1)
    Ap = np.cumsum(Ap)
    return Ap, rows.astype(np.int64), vals


def demo_rhs(n, nrhs=1, complex_=False):
    """Right-hand side of the reference demo: b_i = 1 + i/n (CHOLMOD/Demo/cholmod_dl_demo.c:246-258);
    column j adds j; complex adds i*(n/2 - i)/(3n)."""
    i = np.arange(n, dtype=np.float64)
    B = np.empty((n, nrhs), dtype=np.complex128 if complex_ else np.float64, order="F")
    for j in range(nrhs):
        col = 1.0 + i / n + j
        if complex_:
            col = col + 1j * (n / 2.0 - i) / (3.0 * n)
        B[:, j] = col
    return B


def sym_matvec(prob, X):
    """Y = A X for a lower-stored symmetric/Hermitian problem (host check only)."""
    import scipy.sparse as sp
    n = prob["n"]
    Lw = sp.csc_matrix((prob["Ax"], prob["Ai"], prob["Ap"]), shape=(n, n))
    D = sp.diags(Lw.diagonal())
    A = Lw + Lw.conj().T - D
    return A @ X


def backward_error(prob, X, B):
    """||AX-B||_inf / (||A||_inf ||X||_inf + ||B||_inf)  (cholmod_dl_demo.c:614-617)."""
    import scipy.sparse as sp
    n = prob["n"]
    Lw = sp.csc_matrix((prob["Ax"], prob["Ai"], prob["Ap"]), shape=(n, n))
    D = sp.diags(Lw.diagonal())
    A = (Lw + Lw.conj().T - D).tocsr()
    R = A @ X - B
    anorm = abs(A).sum(axis=1).max()
    return float(np.abs(R).max() / (anorm * np.abs(X).max() + np.abs(B).max()))
