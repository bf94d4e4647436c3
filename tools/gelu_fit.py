#!/usr/bin/env python3
"""Coefficients of the GELU epilogue's normal-CDF polynomial (csrc/common.h: phi_poly2):
Phi(x) - 1/2 = t * Q(t^2), t = clamp(x, -4.5, 4.5) / 4.5, Q of degree 9 (an odd polynomial of degree 19 fitted at Chebyshev
nodes).  Prints the table and the worst error of the fp32 Horner evaluation against scipy's erf."""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf

L, DEG = 4.5, 19
x = np.cos(np.pi * (np.arange(8001) + 0.5) / 8001)
c = C.chebfit(x, 0.5 * erf(x * L / np.sqrt(2)), DEG)
odd = C.cheb2poly(c)[1::2]
print("Q coefficients, highest power first:")
print(", ".join(f"{a:.9e}f" for a in odd[::-1]))
xx = np.linspace(-6, 6, 1200001).astype(np.float32)
t = (np.clip(xx, -L, L) * np.float32(1 / L)).astype(np.float32)
u = (t * t).astype(np.float32)
acc = np.full_like(u, np.float32(odd[-1]))
for a in odd[-2::-1]:
    acc = (acc * u + np.float32(a)).astype(np.float32)
phi = (np.float32(0.5) + t * acc).astype(np.float32)
ref = 0.5 + 0.5 * erf(xx.astype(np.float64) / np.sqrt(2))
print(f"max |Phi_poly - Phi| over [-6, 6] in fp32: {np.abs(phi - ref).max():.3e}")
print(f"max |x| * |Phi_poly - Phi| (error of GELU itself): {(np.abs(xx) * np.abs(phi - ref)).max():.3e}")
