"""GPU: hand-written blocked Cholesky (pxr_dense_spd_solve) vs numpy.linalg on SPD systems."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 7, 64, 65, 130, 500, 1593])
def test_spd_solve_matches_numpy(ctx, n):
    rng = np.random.default_rng(n)
    M = rng.normal(size=(n, n + 5))
    A = M @ M.T + np.eye(n) * 1e-3 * n
    b = rng.normal(size=n)
    x_ref = np.linalg.solve(A, b)
    dA = ctx.to_device(np.triu(A) + np.tril(np.full((n, n), np.nan), -1))   # lower part must be ignored
    db = ctx.to_device(b)
    info = C.c_int(-1)
    from pixsfm_amd._lib import check
    check(ctx.lib.pxr_dense_spd_solve(ctx.handle, dA.ptr, n, db.ptr, C.byref(info)), "pxr_dense_spd_solve")
    assert info.value == 0
    x = db.download()
    assert np.abs(x - x_ref).max() <= 1e-9 * max(1.0, np.abs(x_ref).max()) * np.linalg.cond(A) ** 0.5
    L = np.tril(dA.download().T)        # row-major upper == column-major lower
    assert np.abs(L @ L.T - A).max() < 1e-10 * np.abs(A).max()


@pytest.mark.parametrize("pivot", [70, 71, 0, 63, 64, 99])      # the panel sweeps column PAIRS: first and second of a pair
def test_not_positive_definite_is_reported(ctx, pivot):
    n = 100
    A = np.eye(n); A[pivot, pivot] = -1.0
    dA, db = ctx.to_device(A), ctx.to_device(np.ones(n))
    info = C.c_int(0)
    ctx.lib.pxr_dense_spd_solve(ctx.handle, dA.ptr, n, db.ptr, C.byref(info))
    assert info.value == pivot + 1


def test_pivot_that_turns_negative_only_after_elimination(ctx):
    """positive diagonal, indefinite matrix: the failing pivot is the SECOND column of a pair (d11 - l10^2 <= 0)"""
    n = 40
    A = np.eye(n); A[6, 7] = A[7, 6] = 2.0
    dA, db = ctx.to_device(np.triu(A)), ctx.to_device(np.ones(n))
    info = C.c_int(0)
    ctx.lib.pxr_dense_spd_solve(ctx.handle, dA.ptr, n, db.ptr, C.byref(info))
    assert info.value == 8


def test_more_panel_workgroups_than_compute_units(ctx):
    """n > 48 * 256: the first block steps have more panel workgroups than the GPU has CUs, so some of them start after
    workgroup 0 has finished its tile -- they must still find the tile's INPUT values (the factor of a diagonal tile goes
    to the workspace, not over the tile).  Checked through the residual of the solve."""
    n = 12352
    rng = np.random.default_rng(5)
    M = rng.normal(size=(n, 64))
    A = M @ M.T + np.diag(rng.uniform(50.0, 100.0, n))
    b = rng.normal(size=n)
    info = C.c_int(-1)
    dA, db = ctx.to_device(np.triu(A)), ctx.to_device(b)
    assert ctx.lib.pxr_dense_spd_solve(ctx.handle, dA.ptr, n, db.ptr, C.byref(info)) == 0 and info.value == 0
    x = db.download()
    assert np.linalg.norm(A @ x - b) <= 1e-10 * np.linalg.norm(b)
    L = np.triu(dA.download()).T                      # the returned factor, diagonal tiles included
    rows = rng.integers(0, n, 40)
    assert np.allclose((L[rows] @ L.T), A[rows], rtol=0, atol=1e-9 * np.abs(A).max())


@pytest.mark.parametrize("n", [64, 130, 1593])
def test_single_launch_variant_gives_the_same_factor(ctx, n, monkeypatch):
    """PXR_CHOL_ONE_LAUNCH=1: all block steps in one launch with step counters instead of kernel boundaries (an experiment that
    measured slower than the chain of launches and is off by default; it must still be right)."""
    rng = np.random.default_rng(100 + n)
    M = rng.normal(size=(n, n + 5))
    A = M @ M.T + np.eye(n) * 1e-3 * n
    b = rng.normal(size=n)
    out = []
    for one in (False, True):
        if one:
            monkeypatch.setenv("PXR_CHOL_ONE_LAUNCH", "1")
        dA, db = ctx.to_device(np.triu(A)), ctx.to_device(b)
        info = C.c_int(-1)
        assert ctx.lib.pxr_dense_spd_solve(ctx.handle, dA.ptr, n, db.ptr, C.byref(info)) == 0 and info.value == 0
        out.append((db.download(), dA.download()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(np.triu(out[0][1]), np.triu(out[1][1]))
