#!/usr/bin/env python3
"""Known-answer vectors of the reference's spherical-harmonics kernel (row a6): tests/golden/kat_sh.npz.

    python tests/golden/make_kat_sh.py            (in the build container: reads /root/reference, writes the fixture)

The reference's kernel_sh (encoder/shencoder/src/shencoder.cu:28-357) is CUDA and cannot run here, but its arithmetic is 64 value lines
(`outputs[k] = <polynomial in x, y, z> ;`, :53-122) and 3 x 64 Jacobian lines (`dx[k] = ...`, `dy[k] = ...`, `dz[k] = ...`, :127-356) of plain C
expressions over the locals of :46-50 (xy, xz, yz, x2, y2, z2, xyz, x4, y4, z4, x6, y6, z6).  This script PARSES those lines as data -- the
literal coefficients and monomials -- evaluates them in float64 with Python's own arithmetic on 256 seeded unit vectors, and stores inputs and
expected outputs.  Nothing of the reference's text is stored: the fixture holds numbers only.  An error in OUR generated table
(tools/gen_sh_tables.py -> oracle/ac_sh_table.h == csrc/ac_sh_table.hpp: sign, ordering, coefficient) shows up against these values; the GPU == oracle
tests cannot see one, because both sides use the same table (What's weak 1b)."""
import os
import re
import sys

import numpy as np

SRC = "/root/reference/encoder/shencoder/src/shencoder.cu"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_sh.npz")

LINE = re.compile(r"^\s*(outputs|dx|dy|dz)\[(\d+)\]\s*=\s*(.+?)\s*;")


def to_python(expr):
    expr = re.sub(r"(\d+\.\d*(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)f\b", r"\1", expr)       # 3.0f -> 3.0
    expr = re.sub(r"\bpow\(\s*(\w+)\s*,\s*(\d+)\s*\)", r"(\1**\2)", expr)                  # pow(z, 3) (shencoder.cu:305)
    assert re.fullmatch(r"[-+*/(). \w]*", expr), expr                                       # numbers, the local names, + - * / ( ) and ** only
    return expr


def parse():
    tab = {"outputs": {}, "dx": {}, "dy": {}, "dz": {}}
    for line in open(SRC):
        m = LINE.match(line)
        if m:
            which, k, expr = m.group(1), int(m.group(2)), to_python(m.group(3))
            assert k not in tab[which], (which, k)
            tab[which][k] = compile(expr, f"{which}[{k}]", "eval")
    for which, t in tab.items():
        assert sorted(t) == list(range(64)), (which, len(t))
    return tab


def evaluate(tab, d):
    x, y, z = (d[:, i].astype(np.float64) for i in range(3))
    env = dict(x=x, y=y, z=z, xy=x * y, xz=x * z, yz=y * z, x2=x * x, y2=y * y, z2=z * z, xyz=x * y * z)          # shencoder.cu:46-50
    env.update(x4=env["x2"] ** 2, y4=env["y2"] ** 2, z4=env["z2"] ** 2)
    env.update(x6=env["x4"] * env["x2"], y6=env["y4"] * env["y2"], z6=env["z4"] * env["z2"])
    one = np.ones_like(x)
    ev = lambda code: np.asarray(eval(code, {"__builtins__": {}}, env), np.float64) * one
    val = np.stack([ev(tab["outputs"][k]) for k in range(64)], axis=1)
    jac = np.stack([np.stack([ev(tab[w][k]) for k in range(64)], axis=1) for w in ("dx", "dy", "dz")], axis=1)
    return val, jac


def main():
    tab = parse()
    rs = np.random.RandomState(606)
    d = rs.normal(0, 1, (256, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:6] = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]], np.float64)      # the axes: where many basis functions vanish
    d = d.astype(np.float32)                                   # the inputs the encoders receive are fp32; the expected values are fp64 OF those fp32 inputs
    val, jac = evaluate(tab, d)
    # the basis is orthonormal on the sphere: a Monte-Carlo Gram matrix over 200 000 directions stays within sampling noise of the identity
    # (a guard against a parsing slip -- two lines swapped would still be orthonormal, which is why the VALUES are the fixture)
    big = rs.normal(0, 1, (200000, 3)); big /= np.linalg.norm(big, axis=1, keepdims=True)
    v, _ = evaluate(tab, big)
    gram = 4.0 * np.pi * (v.T @ v) / len(big)
    assert np.abs(gram - np.eye(64)).max() < 0.05, np.abs(gram - np.eye(64)).max()
    np.savez_compressed(OUT, dirs=d, values=val, jacobian=jac,
                        source=np.array("encoder/shencoder/src/shencoder.cu:53-122 (values), :127-356 (dx, dy, dz); parsed as data, evaluated in float64"))
    print("wrote", OUT, val.shape, jac.shape, "max |value|", float(np.abs(val).max()), "max |jac|", float(np.abs(jac).max()))


if __name__ == "__main__":
    sys.exit(main())
