"""Minimax (Lawson-iterated weighted least squares) odd polynomial erf(z) ~ z P(z^2) on |z| <= zmax, evaluated in fp32 with the clamp:
the bf16-mode GELU of csrc/common.h (erf_poly2). python tools/experiments/erf_poly_fit.py"""
import numpy as np
from scipy.special import erf


def lawson(zmax, deg, iters=80, n=6000):
    k = np.arange(n)
    u = (np.cos(np.pi * (k + 0.5) / n) + 1) / 2 * zmax ** 2
    z = np.sqrt(u)
    target = erf(z)
    V = np.vander(u, deg + 1, increasing=True) * z[:, None]
    w = np.ones(n)
    for _ in range(iters):
        c, *_ = np.linalg.lstsq(V * np.sqrt(w)[:, None], target * np.sqrt(w), rcond=None)
        r = np.abs(V @ c - target)
        w = w * (r + 1e-300)
        w /= w.sum()
    return c


def max_err_fp32(c, zmax):
    zz = np.linspace(-zmax * 1.5, zmax * 1.5, 400001).astype(np.float32)
    zc = np.clip(zz, np.float32(-zmax), np.float32(zmax))
    uu = zc * zc
    p = np.zeros_like(uu) + np.float32(c[-1])
    for a in c[-2::-1]:
        p = p * uu + np.float32(a)
    return np.abs((p * zc).astype(np.float64) - erf(zz.astype(np.float64))).max()


if __name__ == '__main__':
    for zmax, deg in ((3.0, 8), (3.2, 8), (3.0, 7)):
        c = lawson(zmax, deg)
        print(zmax, deg, f'max |err| fp32 = {max_err_fp32(c, zmax):.2e}', [float(np.float32(a)) for a in c])
