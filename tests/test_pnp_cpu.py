"""CPU: the PnP math shared with the HIP kernels (csrc/pnp_math.h) built for the host with g++:
quartic roots, Grunert P3P on synthetic poses (incl. near-planar / far / skewed triangles), the
Gauss-Newton refinement.  No GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("pnp") / "libpnp_host.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", out, os.path.join(ROOT, "tests", "pnp_host.cpp")])
    L = ctypes.CDLL(out)
    L.t_quartic.restype = ctypes.c_int
    L.t_quartic.argtypes = [ctypes.c_double] * 4 + [ctypes.POINTER(ctypes.c_double)]
    L.t_p3p.restype = ctypes.c_int
    L.t_refine.restype = ctypes.c_int
    L.t_reproj.restype = ctypes.c_double
    return L


def _ptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def random_pose(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return R


def test_quartic_roots(lib):
    rng = np.random.default_rng(0)
    for _ in range(200):
        r = np.sort(rng.uniform(-3, 3, size=4))
        c = np.poly(r)               # x^4 + b x^3 + ...
        out = np.zeros(4)
        n = lib.t_quartic(c[1], c[2], c[3], c[4], _ptr(out))
        assert n == 4
        assert np.allclose(np.sort(out), r, atol=1e-6)
    # two real + two complex roots
    c = np.poly([1.5, -0.7, 0.3 + 2j, 0.3 - 2j]).real
    out = np.zeros(4)
    n = lib.t_quartic(c[1], c[2], c[3], c[4], _ptr(out))
    assert n == 2 and np.allclose(np.sort(out[:2]), [-0.7, 1.5], atol=1e-8)


@pytest.mark.parametrize("planar", [False, True])
def test_p3p_recovers_pose(lib, planar):
    rng = np.random.default_rng(1 + planar)
    ok = 0
    trials = 300
    for _ in range(trials):
        R = random_pose(rng)
        X = rng.uniform(-0.5, 0.5, size=(3, 3))
        if planar:
            X[:, 2] = 0.0
        t = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(1.5, 4.0)])
        Xc = X @ R.T + t
        if (Xc[:, 2] <= 0.1).any():
            continue
        y = Xc / np.linalg.norm(Xc, axis=1, keepdims=True)
        out = np.zeros(48)
        n = lib.t_p3p(_ptr(np.ascontiguousarray(y)), _ptr(np.ascontiguousarray(X)), _ptr(out))
        assert 1 <= n <= 4
        poses = out[:12 * n].reshape(n, 12)
        errs = [np.abs(p[:9].reshape(3, 3) - R).max() + np.abs(p[9:] - t).max() for p in poses]
        for p in poses:                                # every returned pose reproduces the three bearings
            Rp, tp = p[:9].reshape(3, 3), p[9:]
            assert abs(np.linalg.det(Rp) - 1) < 1e-8 and np.allclose(Rp @ Rp.T, np.eye(3), atol=1e-8)
            Yc = X @ Rp.T + tp
            Yc /= np.linalg.norm(Yc, axis=1, keepdims=True)
            assert np.abs(Yc - y).max() < 1e-3   # near-double roots are ill-conditioned; RANSAC scoring rejects them
        ok += min(errs) < 1e-6
    assert ok >= 0.98 * trials                          # the true pose is among the solutions


def test_gauss_newton_refinement(lib):
    rng = np.random.default_rng(5)
    K4 = np.array([600.0, 610.0, 256.0, 250.0])
    for _ in range(20):
        R = random_pose(rng)
        X = rng.uniform(-0.3, 0.3, size=(60, 3))
        t = np.array([0.05, -0.02, 2.0])
        Xc = X @ R.T + t
        uv = np.stack([K4[0] * Xc[:, 0] / Xc[:, 2] + K4[2], K4[1] * Xc[:, 1] / Xc[:, 2] + K4[3]], 1)
        # start from a perturbed pose (5 degrees / 5 cm)
        w = rng.normal(size=3)
        w *= np.deg2rad(5) / np.linalg.norm(w)
        Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        R0 = (np.eye(3) + Wx + 0.5 * Wx @ Wx) @ R
        u, _, vt = np.linalg.svd(R0)
        R0 = u @ vt
        pose = np.concatenate([R0.reshape(-1), t + rng.normal(size=3) * 0.05])
        rc = lib.t_refine(_ptr(pose), _ptr(K4), _ptr(np.ascontiguousarray(X)), _ptr(np.ascontiguousarray(uv)), 60, 10)
        assert rc == 0
        assert np.abs(pose[:9].reshape(3, 3) - R).max() < 1e-8 and np.abs(pose[9:] - t).max() < 1e-8
