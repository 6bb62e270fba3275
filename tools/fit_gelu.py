#!/usr/bin/env python3
"""Coefficients of gelu_bf16_x4 (ufm_amd/csrc/common.h):  gelu(x) = relu(x) - |x| * 2^-g(min(|x|, 6)),
g = polynomial fit of -log2 Phi(-a) on [0, 6] (Chebyshev nodes), evaluated here in float32 Horner form against the
exact erf GELU in float64.  CPU only (numpy/scipy)."""
import numpy as np
from numpy.polynomial import Polynomial, chebyshev as C
from scipy.special import erf, log_ndtr

A, DEG = 6.0, 6
n = 400
xs = np.cos(np.pi * (np.arange(n) + 0.5) / n) * (A / 2) + A / 2
cheb = C.Chebyshev.fit(xs, -log_ndtr(-xs) / np.log(2.0), DEG, domain=[0, A])
coef = cheb.convert(kind=Polynomial).convert(domain=[-1, 1], window=[-1, 1]).coef
print("coefficients, constant term first:", ", ".join(f"{c:.9g}f" for c in coef))
c32 = coef.astype(np.float32)
x = np.concatenate([np.linspace(-12, 12, 2000001), np.random.default_rng(0).normal(size=1000000) * 2]).astype(np.float32)
a = np.minimum(np.abs(x), np.float32(A))
acc = np.full_like(a, c32[-1])
for c in c32[-2::-1]:
    acc = (acc * a + c).astype(np.float32)
y = (np.maximum(x, 0) - np.abs(x) * np.exp2(-acc).astype(np.float32)).astype(np.float32)
xd = x.astype(np.float64)
ref = 0.5 * xd * (1 + erf(xd / np.sqrt(2)))
err = np.abs(y - ref)
m = np.abs(ref) > 1e-6
print(f"max |err| {err.max():.3g}   max rel err (|y| > 1e-6) {(err[m] / np.abs(ref[m])).max():.3g}   "
      f"max err in bf16 ulps of the result {(err[m] / (np.abs(ref[m]) * 2.0**-8)).max():.3g}")
