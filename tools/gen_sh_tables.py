#!/usr/bin/env python3
"""Generate the real-spherical-harmonics monomial tables (degree 1..8, 64 basis functions)
used by both the CPU oracle (oracle/ac_sh_table.h) and the HIP kernel
(avatarcraft_amd/csrc/ac_sh_table.hpp).

The basis is the Cartesian real-SH family that the reference's SH encoder evaluates on the RAW
input (no normalisation): reference encoder/shencoder/src/shencoder.cu:45-122 (values) and
:127-356 (analytic d/dx, d/dy, d/dz).  Here every basis function is written as
    coefficient * polynomial(x, y, z)
in closed form, expanded symbolically into monomials  c * x^a y^b z^c  and differentiated
symbolically, so values and Jacobian come from one table and cannot drift apart.
Self-check: orthonormality of the 64 functions on the unit sphere (Lebedev-free: Gauss-Legendre
x uniform-phi quadrature).  Evaluation order (both implementations): monomials in table order,
monomial = (x^a * y^b) * z^c with powers by repeated multiplication, acc = fma(c, monomial, acc).
"""
import math, sys, os
from fractions import Fraction

class Poly:
    def __init__(self, terms=None):
        self.t = {k: v for k, v in (terms or {}).items() if v != 0}
    @staticmethod
    def const(c): return Poly({(0, 0, 0): Fraction(c)})
    def _c(self, o): return o if isinstance(o, Poly) else Poly.const(o)
    def __add__(self, o):
        o = self._c(o); r = dict(self.t)
        for k, v in o.t.items(): r[k] = r.get(k, 0) + v
        return Poly(r)
    __radd__ = __add__
    def __neg__(self): return Poly({k: -v for k, v in self.t.items()})
    def __sub__(self, o): return self + (-self._c(o))
    def __rsub__(self, o): return self._c(o) - self
    def __mul__(self, o):
        o = self._c(o); r = {}
        for (a, b, c), v in self.t.items():
            for (d, e, f), w in o.t.items():
                k = (a + d, b + e, c + f); r[k] = r.get(k, 0) + v * w
        return Poly(r)
    __rmul__ = __mul__
    def __pow__(self, n):
        r = Poly.const(1)
        for _ in range(n): r = r * self
        return r
    def diff(self, axis):
        r = {}
        for k, v in self.t.items():
            if k[axis] == 0: continue
            kk = list(k); kk[axis] -= 1
            r[tuple(kk)] = r.get(tuple(kk), 0) + v * k[axis]
        return Poly(r)
    def eval(self, x, y, z):
        return sum(float(v) * x ** a * y ** b * z ** c for (a, b, c), v in self.t.items())

x = Poly({(1, 0, 0): Fraction(1)}); y = Poly({(0, 1, 0): Fraction(1)}); z = Poly({(0, 0, 1): Fraction(1)})
x2, y2, z2 = x * x, y * y, z * z
x4, y4, z4 = x2 * x2, y2 * y2, z2 * z2
x6, y6, z6 = x4 * x2, y4 * y2, z4 * z2
s = math.sqrt; spi = math.sqrt(math.pi)

# (coefficient, polynomial) per basis index, band l = 0..7, m = -l..l
BASIS = [
    (1 / (2 * spi), Poly.const(1)),
    (-s(3) / (2 * spi), y), (s(3) / (2 * spi), z), (-s(3) / (2 * spi), x),
    (s(15) / (2 * spi), x * y), (-s(15) / (2 * spi), y * z), (s(5) / (4 * spi), 3 * z2 - 1),
    (-s(15) / (2 * spi), x * z), (s(15) / (4 * spi), x2 - y2),
    (s(70) / (8 * spi), y * (-3 * x2 + y2)), (s(105) / (2 * spi), x * y * z),
    (s(42) / (8 * spi), y * (1 - 5 * z2)), (s(7) / (4 * spi), z * (5 * z2 - 3)),
    (s(42) / (8 * spi), x * (1 - 5 * z2)), (s(105) / (4 * spi), z * (x2 - y2)),
    (s(70) / (8 * spi), x * (-x2 + 3 * y2)),
    (3 * s(35) / (4 * spi), x * y * (x2 - y2)), (3 * s(70) / (8 * spi), y * z * (-3 * x2 + y2)),
    (3 * s(5) / (4 * spi), x * y * (7 * z2 - 1)), (3 * s(10) / (8 * spi), y * z * (3 - 7 * z2)),
    (3 / (16 * spi), -30 * z2 + 35 * z4 + 3), (3 * s(10) / (8 * spi), x * z * (3 - 7 * z2)),
    (3 * s(5) / (8 * spi), (x2 - y2) * (7 * z2 - 1)), (3 * s(70) / (8 * spi), x * z * (-x2 + 3 * y2)),
    (3 * s(35) / (16 * spi), -6 * x2 * y2 + x4 + y4),
    (3 * s(154) / (32 * spi), y * (10 * x2 * y2 - 5 * x4 - y4)), (3 * s(385) / (4 * spi), x * y * z * (x2 - y2)),
    (-s(770) / (32 * spi), y * (3 * x2 - y2) * (9 * z2 - 1)), (s(1155) / (4 * spi), x * y * z * (3 * z2 - 1)),
    (s(165) / (16 * spi), y * (14 * z2 - 21 * z4 - 1)), (s(11) / (16 * spi), z * (-70 * z2 + 63 * z4 + 15)),
    (s(165) / (16 * spi), x * (14 * z2 - 21 * z4 - 1)), (s(1155) / (8 * spi), z * (x2 - y2) * (3 * z2 - 1)),
    (-s(770) / (32 * spi), x * (x2 - 3 * y2) * (9 * z2 - 1)), (3 * s(385) / (16 * spi), z * (-6 * x2 * y2 + x4 + y4)),
    (3 * s(154) / (32 * spi), x * (10 * x2 * y2 - x4 - 5 * y4)),
    (s(6006) / (32 * spi), x * y * (-10 * x2 * y2 + 3 * x4 + 3 * y4)),
    (3 * s(2002) / (32 * spi), y * z * (10 * x2 * y2 - 5 * x4 - y4)),
    (3 * s(91) / (8 * spi), x * y * (x2 - y2) * (11 * z2 - 1)),
    (-s(2730) / (32 * spi), y * z * (3 * x2 - y2) * (11 * z2 - 3)),
    (s(2730) / (32 * spi), x * y * (-18 * z2 + 33 * z4 + 1)),
    (s(273) / (16 * spi), y * z * (30 * z2 - 33 * z4 - 5)),
    (s(13) / (32 * spi), 105 * z2 - 315 * z4 + 231 * z6 - 5),
    (s(273) / (16 * spi), x * z * (30 * z2 - 33 * z4 - 5)),
    (s(2730) / (64 * spi), (x2 - y2) * (11 * z2 * (3 * z2 - 1) - 7 * z2 + 1)),
    (-s(2730) / (32 * spi), x * z * (x2 - 3 * y2) * (11 * z2 - 3)),
    (3 * s(91) / (32 * spi), (11 * z2 - 1) * (-6 * x2 * y2 + x4 + y4)),
    (3 * s(2002) / (32 * spi), x * z * (10 * x2 * y2 - x4 - 5 * y4)),
    (s(6006) / (64 * spi), 15 * x2 * y4 - 15 * x4 * y2 + x6 - y6),
    (3 * s(715) / (64 * spi), y * (-21 * x2 * y4 + 35 * x4 * y2 - 7 * x6 + y6)),
    (3 * s(10010) / (32 * spi), x * y * z * (-10 * x2 * y2 + 3 * x4 + 3 * y4)),
    (-3 * s(385) / (64 * spi), y * (13 * z2 - 1) * (-10 * x2 * y2 + 5 * x4 + y4)),
    (3 * s(385) / (8 * spi), x * y * z * (x2 - y2) * (13 * z2 - 3)),
    (-3 * s(35) / (64 * spi), y * (3 * x2 - y2) * (13 * z2 * (11 * z2 - 3) - 27 * z2 + 3)),
    (3 * s(70) / (32 * spi), x * y * z * (-110 * z2 + 143 * z4 + 15)),
    (s(105) / (64 * spi), y * (-135 * z2 + 495 * z4 - 429 * z6 + 5)),
    (s(15) / (32 * spi), z * (315 * z2 - 693 * z4 + 429 * z6 - 35)),
    (s(105) / (64 * spi), x * (-135 * z2 + 495 * z4 - 429 * z6 + 5)),
    (s(70) / (64 * spi), z * (x2 - y2) * (143 * z2 * (3 * z2 - 1) - 187 * z2 + 45)),
    (-3 * s(35) / (64 * spi), x * (x2 - 3 * y2) * (13 * z2 * (11 * z2 - 3) - 27 * z2 + 3)),
    (3 * s(385) / (32 * spi), z * (13 * z2 - 3) * (-6 * x2 * y2 + x4 + y4)),
    (-3 * s(385) / (64 * spi), x * (13 * z2 - 1) * (-10 * x2 * y2 + x4 + 5 * y4)),
    (3 * s(10010) / (64 * spi), z * (15 * x2 * y4 - 15 * x4 * y2 + x6 - y6)),
    (3 * s(715) / (64 * spi), x * (-35 * x2 * y4 + 21 * x4 * y2 - x6 + 7 * y6)),
]
assert len(BASIS) == 64

def selfcheck():
    import numpy as np
    nodes, wts = np.polynomial.legendre.leggauss(24)
    nphi = 48
    G = np.zeros((64, 64))
    for ct, w in zip(nodes, wts):
        st = math.sqrt(1 - ct * ct)
        for k in range(nphi):
            ph = 2 * math.pi * (k + 0.5) / nphi
            v = np.array([c * p.eval(st * math.cos(ph), st * math.sin(ph), ct) for c, p in BASIS])
            G += w * (2 * math.pi / nphi) * np.outer(v, v)
    err = np.abs(G - np.eye(64)).max()
    assert err < 1e-9, f"SH basis not orthonormal: {err}"
    return err

def emit(path, guard, qual="static const"):
    kinds = []  # 4 kinds: value, d/dx, d/dy, d/dz
    for axis in (None, 0, 1, 2):
        offs, mono = [0], []
        for c, p in BASIS:
            q = p if axis is None else p.diff(axis)
            for (a, b, cc), v in sorted(q.t.items(), key=lambda kv: (-sum(kv[0]), kv[0])):
                mono.append((c * float(v), a, b, cc))
            offs.append(len(mono))
        kinds.append((offs, mono))
    with open(path, "w") as f:
        f.write("/* GENERATED by tools/gen_sh_tables.py -- do not edit.  Real-SH monomial tables,\n"
                " * 64 basis functions (degree 1..8); kind 0 = value, 1..3 = d/dx, d/dy, d/dz. */\n")
        f.write(f"#ifndef {guard}\n#define {guard}\n")
        for ki, (offs, mono) in enumerate(kinds):
            f.write(f"{qual} unsigned short AC_SH_OFF{ki}[65] = {{{','.join(map(str, offs))}}};\n")
            f.write(f"{qual} float AC_SH_COEF{ki}[{max(1,len(mono))}] = {{\n")
            f.write(",\n".join("  " + ", ".join(f"{float(m[0]).hex()}f" for m in mono[i:i + 4]) for i in range(0, len(mono), 4)))
            f.write("\n};\n")
            # the same coefficients in double (the float ones above are these rounded): the double instantiation of the encoder (shencoder.cu's
            # AT_DISPATCH_FLOATING_TYPES_AND_HALF) evaluates the basis in double like the reference's double literals
            f.write(f"{qual} double AC_SH_COEF{ki}D[{max(1,len(mono))}] = {{\n")
            f.write(",\n".join("  " + ", ".join(f"{float(m[0]).hex()}" for m in mono[i:i + 4]) for i in range(0, len(mono), 4)))
            f.write("\n};\n")
            f.write(f"{qual} unsigned char AC_SH_EXP{ki}[{max(1,len(mono))}][3] = {{\n")
            f.write(",\n".join("  " + ", ".join("{%d,%d,%d}" % m[1:] for m in mono[i:i + 8]) for i in range(0, len(mono), 8)))
            f.write("\n};\n")
        f.write("#endif\n")

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    print("orthonormality max err", selfcheck())
    emit(os.path.join(root, "oracle", "ac_sh_table.h"), "AC_SH_TABLE_ORACLE_H")
    emit(os.path.join(root, "avatarcraft_amd", "csrc", "ac_sh_table.hpp"), "AC_SH_TABLE_HIP_HPP", "static __device__ const")
    print("written")
