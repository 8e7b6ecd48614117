"""mi355_factor_rcond_noalloc (include/mi355_cholmod.h): the allocation-free variant of mi355_factor_rcond -- its
per-workgroup partial results live in a file-scope device array whose per-device address comes from hipGetSymbolAddress.
Checked through the C ABI against mi355_factor_rcond on two calls in a row, the second with a smaller input (a stale
partial result of the first call must not be read), then on a complex factor, a factor that is not positive definite,
and again on the large one."""
import numpy as np
import pytest

from conftest import load_golden, sym_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from suitesparse_amd import engine as e
    assert e.probe() >= 1
    return e


def _factor(engine, kind, size):
    from suitesparse_amd import problems, symbolic
    p = {"lap3d": problems.laplacian_3d, "helm": problems.helmholtz_3d}[kind](size)
    sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], symbolic.nd_grid_order(p["shape"], p["dof"]))
    L = engine.SupernodalFactor(sym, xtype=2 if np.iscomplexobj(p["Ax"]) else 1)
    L.set_matrix_perm(p["Ap"], p["Ai"], p["Ax"], sym["Perm"])
    assert L.numeric() == 0
    return L


def test_rcond_without_allocation_matches_rcond(engine):
    import torch
    out = torch.full((4,), -7.0, dtype=torch.float64, device="cuda")
    big = _factor(engine, "lap3d", 70)          # n = 343 000 > 1024 * 256: the workgroups stride
    small = _factor(engine, "lap3d", 6)         # n = 216: one workgroup, one partial triple
    cplx = _factor(engine, "helm", 8)
    for k, L in enumerate((big, small, cplx, big)):
        want = L.rcond()
        L.rcond_noalloc(out.data_ptr() + 8 * (k % 4))
        got = float(out[k % 4].item())
        assert 0.0 < want <= 1.0
        assert abs(got - want) <= 1e-15 * want, (k, got, want)        # the same reduction: equal to rounding
    # a factorization that stopped at a bad pivot: 0, as cholmod_rcond (L->minor < n)
    g = load_golden("lap3d_8_npd200")
    bad = engine.SupernodalFactor(sym_of(g))
    bad.set_matrix(g["Sp"], g["Si"], g["Sx"])
    assert bad.numeric(beta=g["beta"]) == 1
    bad.rcond_noalloc(out.data_ptr())
    assert float(out[0].item()) == 0.0 == bad.rcond()
    with pytest.raises(engine.Mi355Error):
        bad.rcond_noalloc(0)
    for L in (big, small, cplx, bad):
        L.close()


@pytest.mark.parametrize("name", ["lap3d_8_metis", "helm3d_6_metis", "demo_bcsstk02", "tcov_ex5lo"])
def test_rcond_without_allocation_against_the_reference_definition(engine, name):
    """... and directly against what cholmod_rcond computes from the REFERENCE's factor of the same matrix (the fixture's
    L->x): for an LL' factor (min_j L_jj / max_j L_jj)^2 over the diagonal of every supernode
    (CHOLMOD/Cholesky/cholmod_rcond.c:51-110, the supernodal branch) -- not only against the sibling entry point"""
    import torch
    g = load_golden(name)
    cplx = g["xtype"] == 2
    L = engine.SupernodalFactor(sym_of(g), xtype=2 if cplx else 1)
    L.set_matrix(g["Sp"], g["Si"], g["Sx"])
    assert L.numeric(beta=g["beta"]) == 0
    d = []
    for s in range(g["nsuper"]):
        nscol = int(g["Super"][s + 1] - g["Super"][s]); nsrow = int(g["Lpi"][s + 1] - g["Lpi"][s]); px = int(g["Lpx"][s])
        d.append(np.real(g["Lx"][px + np.arange(nscol) * (nsrow + 1)]))
    d = np.concatenate(d)
    want = (d.min() / d.max()) ** 2
    out = torch.full((1,), -7.0, dtype=torch.float64, device="cuda")
    L.rcond_noalloc(out.data_ptr())
    got = float(out[0].item())
    assert abs(got - want) <= 1e-10 * want, (got, want)
    L.close()
