"""The numerics contract (oracle/det_math.h) pinned to the real functions: each deterministic
sequence must agree with double-precision libm to a few ulp over the ranges the path uses, so that
"bit-exact against the oracle" also means "a faithful sin/cos/log/exp"."""
import numpy as np


def ulp_err(got, want64):
    want32 = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    ulp = np.maximum(ulp, np.finfo(np.float32).tiny)
    return np.abs(got.astype(np.float64) - want64) / ulp


def test_log_accuracy(oracle):
    rng = np.random.default_rng(0)
    x = np.exp(rng.uniform(np.log(1e-38), 0.0, 400_000)).astype(np.float32)
    x = np.concatenate([x, np.float32([1.0, 0.5, 0.70710678, 2.0 ** -32, 1e-38, 1.17549435e-38])])
    got = oracle.math_array(0, x)
    err = ulp_err(got, np.log(x.astype(np.float64)))
    assert err.max() <= 2.0, err.max()
    assert oracle.math_array(0, np.float32([1.0]))[0] == 0.0


def test_sincos_accuracy(oracle):
    u = np.random.default_rng(1).uniform(0, 1, 400_000).astype(np.float32)
    u = np.concatenate([u, np.float32([0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 1.0])])
    for op, fn in ((1, np.sin), (2, np.cos)):
        got = oracle.math_array(op, u).astype(np.float64)
        want = fn(2 * np.pi * u.astype(np.float64))
        # absolute error: GLSL allows 2^-11 on sin/cos; the contract is ~1e-7
        assert np.abs(got - want).max() < 2.5e-7
    s, c = oracle.math_array(1, u).astype(np.float64), oracle.math_array(2, u).astype(np.float64)
    assert np.abs(s * s + c * c - 1).max() < 5e-7
    assert oracle.math_array(1, np.float32([0.25]))[0] == 1.0 and oracle.math_array(2, np.float32([0.5]))[0] == -1.0


def test_exp_accuracy(oracle):
    x = np.random.default_rng(2).uniform(-87, 0, 400_000).astype(np.float32)
    got = oracle.math_array(5, x)
    err = ulp_err(got, np.exp(x.astype(np.float64)))
    assert err.max() <= 2.0, err.max()
    assert oracle.math_array(5, np.float32([0.0, -0.0]))[0] == 1.0
    assert oracle.math_array(5, np.float32([-100.0]))[0] == 0.0


def test_powi_and_f2i(oracle):
    lib = oracle.lib()
    for x in (0.0, 0.5, 0.999, 1.0, 0.99999994):
        want = np.float32(x)
        for _ in range(7):
            want = np.float32(want * want)          # 128 = 2^7: seven squarings
        assert np.float32(lib.oracle_powi(np.float32(x), 128)) == want
    assert lib.oracle_powi(np.float32(3.0), 5) == 243.0 and lib.oracle_powi(np.float32(3.0), 1) == 3.0


def test_sqrt_rcp_are_ieee(oracle):
    x = np.random.default_rng(3).uniform(1e-6, 1e6, 200_000).astype(np.float32)
    assert np.array_equal(oracle.math_array(3, x), np.sqrt(x))
    assert np.array_equal(oracle.math_array(4, x), (np.float32(1.0) / x).astype(np.float32))
