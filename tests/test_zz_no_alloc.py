"""miba_dense_spd_solve_static — the entry point that allocates nothing (include/miba.h): device pointers in and out,
scratch in a file-scope static __device__ array of the library.  Checked through the C ABI against the existing
miba_dense_spd_solve on two calls in a row, the second with a smaller input (the array keeps the first call's contents),
then a larger one again, a matrix that is not positive definite, and a size beyond the static limit.  Bar: both entry
points run the same kernels on the same padded image, so the solutions are bit-identical."""
import numpy as np
import pytest

from bundle_adjustment_amd import miba

import _hip

pytestmark = pytest.mark.gpu


def _spd(n, seed):
    rng = np.random.default_rng(seed)
    B = rng.normal(size=(n, n))
    A = B @ B.T + n * np.eye(n)
    return A, rng.normal(size=n)


def _static_solve(s, A, b):
    """inputs and outputs in device memory that the TEST allocates; the entry point allocates nothing"""
    n = A.shape[0]
    dA, db = _hip.DeviceArray(np.asarray(A, dtype=np.float64)), _hip.DeviceArray(np.asarray(b, dtype=np.float64))
    dx, dst = _hip.DeviceArray(np.full(n, np.nan)), _hip.DeviceArray(np.full(1, -7, dtype=np.int32))
    try:
        s.dense_spd_solve_static(n, dA.ptr, db.ptr, dx.ptr, dst.ptr)
        return dx.get(), int(dst.get()[0])
    finally:
        for d in (dA, db, dx, dst):
            d.free()


def test_static_entry_point_matches_allocating_one():
    s = miba.Solver(device_id=0)
    for n, seed in ((700, 1), (130, 2), (1024, 3), (64, 4), (1, 5)):   # large, then smaller, up to the limit, one block, one entry
        A, b = _spd(n, seed)
        x_ref, st_ref = s.dense_spd_solve(A, b)
        x, st = _static_solve(s, A, b)
        assert st == st_ref == 0
        assert np.array_equal(x, x_ref), (n, np.abs(x - x_ref).max())
        assert np.abs(A @ x - b).max() <= 1e-9 * np.abs(b).max()
    # not positive definite: same status as the allocating entry point; the next call is clean again
    A, b = _spd(200, 6)
    A[150, 150] = -1.0
    _, st_ref = s.dense_spd_solve(A, b)
    _, st = _static_solve(s, A, b)
    assert st == st_ref == 2
    A, b = _spd(90, 7)
    x_ref, _ = s.dense_spd_solve(A, b)
    x, st = _static_solve(s, A, b)
    assert st == 0 and np.array_equal(x, x_ref)
    # beyond the static limit: refused with a clear error, nothing launched
    A, b = _spd(miba.STATIC_SOLVE_MAX_N + 1, 8)
    with pytest.raises(miba.MibaError, match="exceeds the static scratch"):
        _static_solve(s, A, b)
    s.close()


def test_static_entry_point_on_a_second_handle():
    """the device address is looked up once per device and shared by every handle on it"""
    a, b_ = miba.Solver(device_id=0), miba.Solver(device_id=0)
    A, b = _spd(300, 11)
    x1, st1 = _static_solve(a, A, b)
    x2, st2 = _static_solve(b_, A, b)
    a.close(); b_.close()
    assert st1 == st2 == 0 and np.array_equal(x1, x2)
