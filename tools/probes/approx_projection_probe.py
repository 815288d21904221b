#!/usr/bin/env python3
"""Probe (CPU, NumPy; round 6): how far a CHEAP projection chain lands from the reference's exact
chain, on the bench's own rig -- the numbers behind DESIGN.md 10's "analysed, not built".

exact (backprojection.c:11-21, as csrc/sc_project.h keeps it): p = ((R0 x + R1 y) + R2 z) + t per row, every product and
sum rounded to binary32; q = p / p_z correctly rounded; uf = (q fx) + cx; the pixel is (int)uf.
cheap: b = (R0 x + R1 y) + t once per column and view, p~ = fma(R2, z, b), r = 1 / p~_z (a reciprocal good to 1 ulp; here
the correctly rounded one, so 1 ulp of slack is ADDED to the bound below), uf~ = fma(p~ r, fx, cx): 3 + 1 + 4 vector
instructions where the exact chain has 22.

Reports, over N sampled (voxel, view) pairs of the 512^3 x 72 rig: the largest and the quantiles of |uf - uf~| (pixels),
the first-order bound of that difference evaluated per pair (the probe checks bound >= difference everywhere), the
share of pairs inside the guard band (within `bound` of an integer in u or v), and how many pairs OUTSIDE the band
disagree on the pixel (must be 0).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from plant3dvision_amd import scenes  # noqa: E402

F = np.float32
U = 2.0 ** -24  # unit round-off of binary32


def fma(a, b, c):
    # a * b is exact in binary64 (24 + 24 bits); the sum is rounded to binary64, then to binary32: a double rounding that
    # differs from a true FMA in ~2^-29 of the cases by one ulp -- inside the 1 ulp of slack the bound carries for the
    # reciprocal
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    shape, origin, vs, views = scenes.make_scene(512, 72, "empty")
    ox, oy, oz = (F(v) for v in origin)
    vs = F(vs)
    worst = 0.0
    worst_ratio = 0.0
    diffs = []
    in_band = 0
    disagree_outside = 0
    disagree_all = 0
    per = n // len(views)
    for K, R, t, m in views:
        K, R, t = K.astype(F), R.astype(F).reshape(9), t.astype(F)
        H, W = m.shape
        i = rng.integers(0, shape[0], per)
        j = rng.integers(0, shape[1], per)
        k = rng.integers(0, shape[2], per)
        x = ox + i.astype(F) * vs
        y = oy + j.astype(F) * vs
        z = oz + k.astype(F) * vs
        res_e, res_a, bounds = [], [], []
        pz_e = pz_a = None
        rows = []
        for row, tt in ((6, t[2]), (0, t[0]), (3, t[1])):
            a = R[row] * x + R[row + 1] * y                      # the hoisted partial sum (shared by both chains)
            c = R[row + 2] * z
            pe = (a + c) + tt                                      # exact chain
            pa = fma(np.full(per, R[row + 2], F), z, a + tt)      # cheap chain
            # first-order bound of |pe - pa|: the roundings of c, a + c, (a + c) + t on one side, of a + t and the FMA on the other
            e = U * (np.abs(c) + np.abs(a + c) + np.abs(pe) + np.abs(a + tt) + np.abs(pa)).astype(np.float64)
            rows.append((pe, pa, e))
        (pze, pza, ez), (pxe, pxa, ex), (pye, pya, ey) = rows
        out = []
        for (pe, pa, e), f, c in (((pxe, pxa, ex), K[0], K[2]), ((pye, pya, ey), K[1], K[3])):
            qe = pe / pze                                          # correctly rounded
            ufe = qe * f + c
            r = F(1.0) / pza
            qa = pa * r
            ufa = fma(qa, np.full(per, f, F), np.full(per, c, F))
            q64 = np.abs(pe.astype(np.float64) / pze.astype(np.float64))
            dq = (e + q64 * ez) / np.abs(pze.astype(np.float64)) + 6.0 * U * q64  # operands + division, reciprocal (2 ulp), product
            bound = abs(float(f)) * dq + U * (np.abs(qe * f).astype(np.float64) + np.abs(ufe).astype(np.float64) + np.abs(ufa).astype(np.float64))
            bound *= 1.25                                          # second-order terms
            out.append((ufe, ufa, bound))
        (ue, ua, bu), (ve, va, bv) = out
        d = np.maximum(np.abs(ue.astype(np.float64) - ua), np.abs(ve.astype(np.float64) - va))
        ratio = np.maximum(np.abs(ue.astype(np.float64) - ua) / bu, np.abs(ve.astype(np.float64) - va) / bv)
        worst = max(worst, float(d.max()))
        worst_ratio = max(worst_ratio, float(ratio.max()))
        diffs.append(d[:: max(1, per // 4096)])
        fu = ua.astype(np.float64) - np.floor(ua.astype(np.float64))
        fv = va.astype(np.float64) - np.floor(va.astype(np.float64))
        band = (np.minimum(fu, 1 - fu) < bu) | (np.minimum(fv, 1 - fv) < bv)
        in_band += int(band.sum())
        same = (np.trunc(ue) == np.trunc(ua)) & (np.trunc(ve) == np.trunc(va))
        disagree_all += int((~same).sum())
        disagree_outside += int((~same & ~band).sum())
    d = np.concatenate(diffs)
    total = per * len(views)
    print(f"{total} (voxel, view) pairs of the 512^3 x 72 rig")
    print(f"|uf - uf~| (pixels): max {worst:.3e}, median {np.median(d):.3e}, 99 % {np.quantile(d, 0.99):.3e}, 99.99 % {np.quantile(d, 0.9999):.3e}")
    print(f"largest difference / its first-order bound: {worst_ratio:.3f} (must stay below 1)")
    print(f"pairs inside the guard band: {in_band} = {100.0 * in_band / total:.3f} %")
    print(f"pairs whose pixel differs between the chains: {disagree_all} = {100.0 * disagree_all / total:.4f} %; of them outside the band: {disagree_outside} (must be 0)")
    for lanes in (128, 256, 512):
        print(f"  a wavefront turn of {lanes} pairs holds a guarded pair with probability {100.0 * (1 - (1 - in_band / total) ** lanes):.1f} %")


if __name__ == "__main__":
    main()
