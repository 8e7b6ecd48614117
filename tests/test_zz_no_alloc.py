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
