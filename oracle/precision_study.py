"""Precision study: which MFMA operand formats keep the class scores within 1e-4 of the reference.

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.  numpy emulation (no GPU):

    python -m oracle.precision_study [--windows 48] [--seeds 42 43] [--out profiles/history/r02_precision_study.json]

The fused kernel (genomad_amd/csrc/gnn_fused.hip) evaluates four contractions per window on the matrix
pipe — conv2, conv3 (igloo.py:66: K = 6·128, N = 128) and y @ w_v of both IGLOO heads (igloo.py:208:
K = 128) — with f32 accumulation.  Each operand is an f32 value that has to be fed to the MFMA as one
or more low-precision "limbs".  This script re-runs the oracle forward pass with those four
contractions replaced by an emulation of a limb scheme (operands rounded exactly as the hardware
formats round, products and sums in float64 so that only the operand formats are measured), and
reports max |Δscore| against the fp64 oracle over the first ``--windows`` synthetic windows, for
every weight seed given.  Everything else (conv1 gather, pair products, softmax, dense head) stays f64,
except that the IGLOO pair products read the activations as the scheme stores them in LDS.

Cost model ("passes"): one v_mfma_f32_32x32x16_{bf16,f16} pass over the K range = 1.0; i8 MFMA = 0.5;
a 32-element K block of the MX-scaled v_mfma_scale_f32_32x32x64_f8f6f4 = 0.5 in fp8 (e4m3/e5m2; the schemes
below use two blocks = one instruction per 32 channels = 1.0) and 0.25 in fp6/fp4
(/opt/skills/guides/MI355X_MICROARCH.md "Matrix cores": 2382 / 4404 / 4686 / 8939 / 9099 TFLOP/s measured).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from genomad_amd import synthetic  # noqa: E402
from oracle import igloo_oracle as IO  # noqa: E402
from oracle import sequence_oracle  # noqa: E402

# ------------------------------------------------------------------ hardware number formats (RNE)
FORMATS = {                      # explicit mantissa bits, min normal exponent, largest finite value
    "bf16": (7, -126, 3.3895313892515355e38),
    "fp16": (10, -14, 65504.0),
    "e4m3": (3, -6, 448.0),      # OCP e4m3fn (gfx950: not MI300's fnuz)
    "e5m2": (2, -14, 57344.0),
    "e2m3": (3, 0, 7.5),         # MX fp6
    "e3m2": (2, -2, 28.0),       # MX bf6
    "e2m1": (1, 0, 6.0),         # MX fp4
}
EMAX = {"e4m3": 8, "e5m2": 15, "e2m3": 2, "e3m2": 4, "e2m1": 2}


def rnd(x, fmt):
    """Round to ``fmt``: round-to-nearest-even, gradual underflow, saturating."""
    mant, emin, vmax = FORMATS[fmt]
    x = np.asarray(x, dtype=np.float64)
    ax = np.abs(x)
    e = np.floor(np.log2(np.where(ax > 0, ax, 1.0)))
    e = np.maximum(e, emin)
    q = np.exp2(e - mant)
    return np.clip(np.rint(x / q) * q, -vmax, vmax)


def rnd_mx(x, fmt, axis, block=32):
    """OCP MX block format: ``block`` consecutive elements along ``axis`` share a power-of-two scale
    2^(floor(log2(max|x|)) - emax_elem); elements are rounded to ``fmt`` after scaling (saturating)."""
    x = np.moveaxis(np.asarray(x, dtype=np.float64), axis, -1)
    shp = x.shape
    xb = x.reshape(shp[:-1] + (shp[-1] // block, block))
    amax = np.abs(xb).max(axis=-1, keepdims=True)
    se = np.floor(np.log2(np.where(amax > 0, amax, 1.0))) - EMAX[fmt]
    se = np.clip(se, -127, 127)
    s = np.exp2(se)
    out = (rnd(xb / s, fmt) * s).reshape(shp)
    return np.moveaxis(out, -1, axis)


def limbs(x, fmts):
    """x ≈ sum of limbs, limb i = round_{fmts[i]}(x - previous limbs)."""
    out, r = [], np.asarray(x, dtype=np.float64)
    for f in fmts:
        l = rnd(r, f)
        out.append(l)
        r = r - l
    return out


def int_limbs(x, n, axis_scale):
    """Fixed point on the i8 MFMA: x / s rounded to an integer of n balanced 7-bit digits (s chosen so
    that max|x| over ``axis_scale`` maps to 127·2^(7(n-1)), i.e. the top digit fits int8).  Returns the
    digits already multiplied by their weights, so that they sum to the rounded value."""
    x = np.asarray(x, dtype=np.float64)
    amax = np.abs(x).max(axis=axis_scale, keepdims=True)
    s = np.where(amax > 0, amax, 1.0) / (127.0 * 2.0 ** (7 * (n - 1)))
    q = np.rint(x / s)
    out = []
    for i in range(n):
        w = 2.0 ** (7 * (n - 1 - i))
        d = np.rint(q / w)
        out.append(d * w * s)
        q = q - d * w
    return out


# ------------------------------------------------------------------ limb schemes
# A scheme turns (x rows, w matrix) into a list of (x_limb, w_limb) products to be summed, and says
# how the activations are stored (what the pair-product path reads back).
class Scheme:
    def __init__(self, name, cost, xsplit, wsplit, terms, note="", stored=None, weff_fmt=None):
        self.name, self.cost, self.xsplit, self.wsplit, self.terms, self.note = name, cost, xsplit, wsplit, terms, note
        self.weff_fmt = weff_fmt     # format the folded IGLOO weights (w_mult * w_summer) are rounded to for the pair products (None = f32)
        # what the pair-product path reads back from LDS: by default the sum of all activation limbs; the
        # correction schemes keep [hi, coarse image of x, image of the residual] and read hi + residual image
        self.stored = stored or (lambda limbs: sum(l for l in limbs if l is not None))


def _float_scheme(name, fmt, terms, cost, note=""):
    return Scheme(name, cost, lambda x: limbs(x, [fmt, fmt]), lambda w: limbs(w, [fmt, fmt]), terms, note)


def _corr_x(lo_fmt, mx):
    """activations: [fp16 hi, fp8/fp6 image of x, fp8/fp6 image of the fp16 residual]"""
    def f(x):
        hi = rnd(x, "fp16")
        if mx:      # block-scaled along channels (32 consecutive channels of one row = one MFMA K block)
            return [hi, rnd_mx(x, lo_fmt, axis=-1), rnd_mx(x - hi, lo_fmt, axis=-1)]
        # e4m3 with uniform hardware scales: x as is (range 2^-9 .. 448), residual scaled by 2^11
        return [hi, rnd(x, lo_fmt), rnd((x - hi) * 2048.0, lo_fmt) / 2048.0]
    return f


def _corr_x_fp6_derived(x):
    """f16c6 variant that was NOT built: the residual's block scale derived from the block maximum of x (2^-11 below the x
    image's) instead of from the residuals' own maximum — one amax less per (row, block) in the kernel"""
    x = np.asarray(x, dtype=np.float64)
    hi = rnd(x, "fp16")
    shp = x.shape
    xb = x.reshape(shp[:-1] + (shp[-1] // 32, 32))
    amax = np.abs(xb).max(axis=-1, keepdims=True)
    e = np.floor(np.log2(np.where(amax > 0, amax, 1.0)))
    s, sl = np.exp2(e - 2), np.exp2(e - 13)
    x6 = (rnd(xb / s, "e2m3") * s).reshape(shp)
    xl6 = (rnd((x - hi).reshape(xb.shape) / sl, "e2m3") * sl).reshape(shp)
    return [hi, x6, xl6]


_HI_PLUS_RESIDUAL = lambda l: l[0] + l[2]   # noqa: E731


def _corr_w(lo_fmt):
    """weights (static): [fp16 hi, MX image of w, MX image of the fp16 residual], blocks along K"""
    def f(w):
        hi = rnd(w, "fp16")
        return [hi, rnd_mx(w, lo_fmt, axis=0), rnd_mx(w - hi, lo_fmt, axis=0)]
    return f


def _corr_w_best(w):
    """f16(w) may be either neighbour of w: pick, per weight, the one whose residual the MX-e4m3 image represents
    with the smaller error (block scales from the round-to-nearest residuals, so the choice is local)."""
    w = np.asarray(w, dtype=np.float64)
    near = rnd(w, "fp16")
    ulp = np.exp2(np.maximum(np.floor(np.log2(np.maximum(np.abs(near), 1e-300))), -14) - 10)
    other = near + np.where(w >= near, ulp, -ulp)
    best_h, best_l, best_e = near, None, None
    for cand in (near, other):
        res = w - cand
        img = rnd_mx(res, "e4m3", axis=0)
        err = np.abs(res - img)
        if best_e is None:
            best_h, best_l, best_e = cand.copy(), img.copy(), err
        else:
            take = err < best_e
            best_h = np.where(take, cand, best_h)
            best_l = np.where(take, img, best_l)
            best_e = np.minimum(err, best_e)
    # one consistent MX image of the final residuals (block scales follow the chosen residuals)
    return [best_h, rnd_mx(w, "e4m3", axis=0), rnd_mx(w - best_h, "e4m3", axis=0)]


HH, HL, LH, LL = (0, 0), (0, 1), (1, 0), (1, 1)
SCHEMES = [
    Scheme("f32 (exact operands)", 16.0, lambda x: [x], lambda w: [w], [HH]),
    _float_scheme("bf16x1", "bf16", [HH], 1.0),
    _float_scheme("bf16x2 (hh+lh: x 2 limbs, w 1)", "bf16", [HH, LH], 2.0),
    _float_scheme("bf16x2 (hh+hl: x 1 limb, w 2)", "bf16", [HH, HL], 2.0),
    _float_scheme("bf16x3 (round 1)", "bf16", [HH, HL, LH], 3.0),
    _float_scheme("fp16x1", "fp16", [HH], 1.0),
    _float_scheme("fp16x2 (hh+lh: x 2 limbs, w 1)", "fp16", [HH, LH], 2.0),
    _float_scheme("fp16x2 (hh+hl: x 1 limb, w 2)", "fp16", [HH, HL], 2.0),
    _float_scheme("fp16x3", "fp16", [HH, HL, LH], 3.0),
    Scheme("fp16 + e4m3 corrections (x8*wl8 + xl8*w8)", 2.0, _corr_x("e4m3", False), _corr_w("e4m3"),
           [(0, 0), (1, 2), (2, 1)], "MX fp8 MFMA, K-concatenated: one 32x32x64 per 32 channels", _HI_PLUS_RESIDUAL),
    Scheme("fp16 + e5m2 corrections", 2.0, _corr_x("e5m2", False), _corr_w("e5m2"), [(0, 0), (1, 2), (2, 1)], "", _HI_PLUS_RESIDUAL),
    Scheme("fp16 + e2m3 (fp6, MX both sides) corrections", 1.5, _corr_x("e2m3", True), _corr_w("e2m3"),
           [(0, 0), (1, 2), (2, 1)], "", _HI_PLUS_RESIDUAL),
    Scheme("fp16 + e2m3 corrections, residual scale derived from the x scale", 1.5, _corr_x_fp6_derived, _corr_w("e2m3"),
           [(0, 0), (1, 2), (2, 1)], "f16c6 variant, not built", _HI_PLUS_RESIDUAL),
    Scheme("fp16 + e2m3 corrections, pair products read f16(x) only", 1.5, _corr_x("e2m3", True), _corr_w("e2m3"),
           [(0, 0), (1, 2), (2, 1)], "f16c6 variant, not built", lambda l: l[0]),
    Scheme("fp16 + e2m3 corrections, folded IGLOO weights in f16", 1.5, _corr_x("e2m3", True), _corr_w("e2m3"),
           [(0, 0), (1, 2), (2, 1)], "f16c6 variant, not built", _HI_PLUS_RESIDUAL, weff_fmt="fp16"),
    # error-source split of f16c6: one of the four 4-bit images exact at a time (and all four: what the dropped xl*wl term costs)
    Scheme("fp16 + e2m3 corrections, x image exact", 1.5, lambda x: [rnd(x, "fp16"), np.asarray(x, np.float64), rnd_mx(x - rnd(x, "fp16"), "e2m3", -1)],
           _corr_w("e2m3"), [(0, 0), (1, 2), (2, 1)], "diagnostic", _HI_PLUS_RESIDUAL),
    Scheme("fp16 + e2m3 corrections, x residual exact", 1.5, lambda x: [rnd(x, "fp16"), rnd_mx(x, "e2m3", -1), x - rnd(x, "fp16")],
           _corr_w("e2m3"), [(0, 0), (1, 2), (2, 1)], "diagnostic", _HI_PLUS_RESIDUAL),
    Scheme("fp16 + e2m3 corrections, w image exact", 1.5, _corr_x("e2m3", True),
           lambda w: [rnd(w, "fp16"), np.asarray(w, np.float64), rnd_mx(w - rnd(w, "fp16"), "e2m3", 0)], [(0, 0), (1, 2), (2, 1)], "diagnostic", _HI_PLUS_RESIDUAL),
    Scheme("fp16 + e2m3 corrections, w residual exact", 1.5, _corr_x("e2m3", True),
           lambda w: [rnd(w, "fp16"), rnd_mx(w, "e2m3", 0), w - rnd(w, "fp16")], [(0, 0), (1, 2), (2, 1)], "diagnostic", _HI_PLUS_RESIDUAL),
    Scheme("fp16 + e2m3 corrections, all four images exact (only xl*wl dropped)", 1.5,
           lambda x: [rnd(x, "fp16"), np.asarray(x, np.float64), x - rnd(x, "fp16")],
           lambda w: [rnd(w, "fp16"), np.asarray(w, np.float64), w - rnd(w, "fp16")], [(0, 0), (1, 2), (2, 1)], "diagnostic", _HI_PLUS_RESIDUAL),
    Scheme("fp16 + e3m2 (bf6, MX both sides) corrections", 1.5, _corr_x("e3m2", True), _corr_w("e3m2"),
           [(0, 0), (1, 2), (2, 1)], "", _HI_PLUS_RESIDUAL),
    Scheme("fp16 + e2m1 (fp4, MX both sides) corrections", 1.5, _corr_x("e2m1", True), _corr_w("e2m1"),
           [(0, 0), (1, 2), (2, 1)], "", _HI_PLUS_RESIDUAL),
    Scheme("fp16 + e4m3 correction of x only (xl8*w8)", 1.5, _corr_x("e4m3", False), _corr_w("e4m3"),
           [(0, 0), (2, 1)], "w single fp16", _HI_PLUS_RESIDUAL),
    Scheme("fp16 + e4m3 correction of w only (x8*wl8)", 1.5, _corr_x("e4m3", False), _corr_w("e4m3"),
           [(0, 0), (1, 2)], "x single fp16", _HI_PLUS_RESIDUAL),
    Scheme("bf16 + e4m3 corrections", 2.0,
           lambda x: [rnd(x, "bf16"), rnd(x, "e4m3"), rnd((x - rnd(x, "bf16")) * 256.0, "e4m3") / 256.0],
           lambda w: [rnd(w, "bf16"), rnd_mx(w, "e4m3", 0), rnd_mx(w - rnd(w, "bf16"), "e4m3", 0)],
           [(0, 0), (1, 2), (2, 1)], "", _HI_PLUS_RESIDUAL),
    # 2.5-pass variants: one of the two residual terms in a second f16 pass (1.0; no scaling needed: f16 holds the
    # residual directly, in its subnormal range for small weights), the other one in ONE fp8 K block (0.5)
    Scheme("fp16 x (w 2 limbs) + e4m3 correction of x (xh*wh + xh*wl16 + xl8*w8)", 2.5,
           lambda x: [rnd(x, "fp16"), None, rnd((x - rnd(x, "fp16")) * 2048.0, "e4m3") / 2048.0],
           lambda w: [rnd(w, "fp16"), rnd_mx(w, "e4m3", 0), rnd(w - rnd(w, "fp16"), "fp16")],
           [(0, 0), (0, 2), (2, 1)], "", _HI_PLUS_RESIDUAL),
    Scheme("fp16 x (x 2 limbs) + e4m3 correction of w (xh*wh + xl16*wh + x8*wl8)", 2.5,
           lambda x: [rnd(x, "fp16"), rnd(x, "e4m3"), rnd(x - rnd(x, "fp16"), "fp16")],
           lambda w: [rnd(w, "fp16"), None, rnd_mx(w - rnd(w, "fp16"), "e4m3", 0)],
           [(0, 0), (2, 0), (1, 2)], "", _HI_PLUS_RESIDUAL),
    Scheme("fp16 + e4m3 corrections, weights rounded to the f16 neighbour with the better e4m3 residual", 2.0,
           _corr_x("e4m3", False), lambda w: _corr_w_best(w), [(0, 0), (1, 2), (2, 1)],
           "free at run time: the choice is made when the weights are packed", _HI_PLUS_RESIDUAL),
    # fixed point on the i8 MFMA: x scaled per window-step tile is not emulated here; per-row scale for x
    # (optimistic: the accumulator cannot mix row scales across conv taps) and per-column scale for w
    Scheme("int8 limbs x2/w2, 3 products (optimistic per-row x scale)", 1.5,
           lambda x: int_limbs(x, 2, -1), lambda w: int_limbs(w, 2, 0), [HH, HL, LH]),
    Scheme("int8 limbs x3/w2, 5 products (optimistic per-row x scale)", 2.5,
           lambda x: int_limbs(x, 3, -1), lambda w: int_limbs(w, 2, 0), [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0)]),
    Scheme("int8 limbs x3/w3, 6 products (optimistic per-row x scale)", 3.0,
           lambda x: int_limbs(x, 3, -1), lambda w: int_limbs(w, 3, 0),
           [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (0, 2)]),
]


# ------------------------------------------------------------------ Toom-Cook minimal filtering over time
# conv2 / conv3 only (y @ w_v stays direct f16x3).  Unlike the limb schemes above these rows ALSO emulate the f32 accumulator of
# the MFMA (rounded after every instruction) and the f32 transforms, because the transforms' cancellation is where the error of
# minimal filtering comes from; "direct f16x3, f32 accumulator emulated" is the like-for-like baseline.  oracle/toomcook.py.
class ConvScheme(Scheme):
    def __init__(self, name, cost, conv_fn, note=""):
        base = BY_NAME_EARLY["fp16x3"]
        super().__init__(name, cost, base.xsplit, base.wsplit, base.terms, note)
        self.conv_fn = conv_fn            # (x (B,T,C), kernel (6,C,N), bias, stats) -> pre-activation (B,T,N)


BY_NAME_EARLY = {s.name: s for s in SCHEMES}
TC_STATS = {}


def _tc_conv(m):
    from oracle import toomcook
    packs = {}

    def fn(x, kernel, bias, key):
        if key not in packs:
            packs[key] = toomcook.pack_weights(kernel, m)
            TC_STATS.setdefault(f"F({m},6)", {})[f"max_U {key}"] = [float(v) for v in packs[key]["max_U"]]
        return toomcook.conv_emulated(x, packs[key], bias, stats=TC_STATS.setdefault(f"F({m},6)", {}))
    return fn


def _direct_acc_conv(x, kernel, bias, key):
    from oracle import toomcook
    return toomcook.conv_direct_x3(x, kernel, bias)


SCHEMES += [
    ConvScheme("direct f16x3, f32 accumulator emulated", 3.0, _direct_acc_conv, "today's kernel, like-for-like baseline of the Toom-Cook rows"),
    ConvScheme("toomcook F(2,6) on f16x3 limbs", 3.0 * 7 / 12, _tc_conv(2), "points 0, +-1, +-1/2, 2, inf"),
    ConvScheme("toomcook F(3,6) on f16x3 limbs", 3.0 * 8 / 18, _tc_conv(3), "points 0, +-1, +-2, +-1/2, inf"),
    ConvScheme("toomcook F(4,6) on f16x3 limbs", 3.0 * 9 / 24, _tc_conv(4), "points 0, +-1, +-2, +-1/2, 1/4, inf"),
]


# ------------------------------------------------------------------ f16 hi*hi + int8 cross terms
# "one arithmetic not in this study": the hi*hi product on the f16 MFMA, BOTH cross products on v_mfma_i32_32x32x32_i8 (twice the
# f16 rate) with block-scaled int8 images: 1 + 2 x 0.5 = 2.0 pass equivalents per product instead of 3.0.  conv2 / conv3 in the
# Toom-Cook domain (scales per (point, tile) and per (point, output channel): no conv taps to mix, K = 128), y @ w_v direct with
# scales per row and per column.  What the LDS would hold for the pair products: the f16 hi + lo limbs as today.
def _i8_x(x):
    from oracle import toomcook
    x = np.asarray(x, np.float64)
    hi = rnd(x, "fp16")
    q, a = toomcook.q8_rows(x, axis=-1)
    ql, _ = toomcook.q8_rows(x - hi, axis=-1, tied_to=a)
    return [hi, q * a, ql * a * 2.0 ** -11, rnd(x - hi, "fp16")]


def _i8_w(w):
    from oracle import toomcook
    w = np.asarray(w, np.float64)
    hi = rnd(w, "fp16")
    q, b = toomcook.q8_rows(w, axis=0)
    ql, _ = toomcook.q8_rows(w - hi, axis=0, tied_to=b)
    return [hi, q * b, ql * b * 2.0 ** -11]


class ConvSchemeI8(Scheme):
    def __init__(self, name, cost, conv_fn, note=""):
        super().__init__(name, cost, _i8_x, _i8_w, [(0, 0), (1, 2), (2, 1)], note, stored=lambda l: l[0] + l[3])
        self.conv_fn = conv_fn


def _tc_conv_i8(m):
    from oracle import toomcook
    packs = {}

    def fn(x, kernel, bias, key):
        if key not in packs:
            packs[key] = toomcook.pack_weights(kernel, m)
        return toomcook.conv_emulated_i8cross(x, packs[key], bias, stats=TC_STATS.setdefault(f"F({m},6) i8 cross", {}))
    return fn


SCHEMES += [
    ConvSchemeI8("toomcook F(3,6): f16 hi*hi + int8 x int8 cross terms (block scaled); w_v the same, direct", 2.0 * (0.8535 * 8 / 18 + 0.1465) / (0.8535 * 8 / 18 + 0.1465) * 1.0,
                 _tc_conv_i8(3), "2.0 pass equivalents per product; needs a second (i32) accumulator set beside Toom-Cook's 128 f32 registers"),
]


# ------------------------------------------------------------------ shorter low limbs (round 5 energy probe: MFMA power depends on the operand bits)
def _masked_split(bits):
    from oracle import toomcook
    return lambda v: [rnd(v, "fp16"), toomcook.mask16(rnd(np.asarray(v, np.float64) - rnd(v, "fp16"), "fp16"), bits)]


class ConvSchemeMasked(Scheme):
    def __init__(self, bits):
        super().__init__(f"toomcook F(3,6) on f16x3 limbs, low limbs cut by {bits} mantissa bits (weights and activations, w_v too)", 3.0 * 8 / 18,
                         _masked_split(bits), _masked_split(bits), [HH, HL, LH], "energy probe: same MFMAs, fewer operand bits toggling")
        from oracle import toomcook
        packs = {}

        def fn(x, kernel, bias, key):
            if key not in packs:
                packs[key] = toomcook.pack_weights(kernel, 3)
            return toomcook.conv_emulated(x, packs[key], bias, lo_mask_bits=bits)
        self.conv_fn = fn


SCHEMES += [ConvSchemeMasked(b) for b in (3, 4, 5, 6)]
BY_NAME = {s.name: s for s in SCHEMES}
EXACT = SCHEMES[0]


def contract(xl, wl, terms, taps):
    """sum over limb products of the causal conv (taps=6) or the plain product (taps=1); xl: limbs of
    x (B,T,C), wl: limbs of w (taps*C, N)."""
    out = 0.0
    B, T, C = xl[0].shape
    for (i, j) in terms:
        x, w = xl[i], wl[j]
        if taps == 1:
            out = out + x @ w
        else:
            xp = np.concatenate([np.zeros((B, taps - 1, C)), x], axis=1)
            acc = 0.0
            for k in range(taps):
                acc = acc + xp[:, k:k + T] @ w[k * C:(k + 1) * C]
            out = out + acc
    return out


def forward(tokens, W, layer_scheme):
    """fp64 forward with the four matrix-pipe contractions emulated; ``layer_scheme`` maps
    conv2 / conv3 / wvA / wvB to a Scheme."""
    w = {k: (np.asarray(v, dtype=np.float64) if np.asarray(v).dtype.kind == "f" else np.asarray(v))
         for k, v in W.items()}
    f32 = lambda a: a.astype(np.float32).astype(np.float64)   # noqa: E731  activations are f32 before the split
    x1 = f32(IO._lrelu(IO.conv1_gather(tokens, w["conv1_kernel"], w["conv1_bias"])))

    def stored(x, sch):            # what the LDS holds = what the pair products read back
        return sch.stored(sch.xsplit(x))

    def conv(x, name, sch):
        if hasattr(sch, "conv_fn"):
            return f32(IO._lrelu(sch.conv_fn(x, w[f"{name}_kernel"], w[f"{name}_bias"], (name, id(W)))))
        xl, wl = sch.xsplit(x), sch.wsplit(w[f"{name}_kernel"].reshape(6 * 128, 128))
        return f32(IO._lrelu(contract(xl, wl, sch.terms, 6) + w[f"{name}_bias"]))

    def head(x, hname, sch_v, sch_store):
        P = w[f"{hname}_patches"]
        weff = w[f"{hname}_w_mult"][0] * w[f"{hname}_w_summer"][0, :, 0].reshape(4, 128)[None]
        if sch_store.weff_fmt:
            weff = rnd(weff.astype(np.float32).astype(np.float64), sch_store.weff_fmt)
        xs = stored(x, sch_store)
        m = np.einsum("bpjc,pjc->bp", xs[:, P[:, :, 0], :], weff) + w[f"{hname}_w_bias"]
        alpha = IO._softmax(m @ w[f"{hname}_w_qk"])
        yp = contract(sch_v.xsplit(x), sch_v.wsplit(w[f"{hname}_w_v"][0]), sch_v.terms, 1)
        Tp = x.shape[1] // 8
        yp = f32(yp[:, :Tp * 8].reshape(x.shape[0], Tp, 8, 128).max(axis=2))
        return np.einsum("bq,bqc->bc", alpha, yp)

    x2 = conv(x1, "conv2", layer_scheme["conv2"])
    x3 = conv(x2, "conv3", layer_scheme["conv3"])
    # x1 is stored once: the scheme of its first consumer on the matrix pipe decides its LDS format
    fA = head(x1, "iglooA", layer_scheme["wvA"], layer_scheme["conv2"])
    fB = head(x3, "iglooB", layer_scheme["wvB"], layer_scheme["wvB"])
    f = np.concatenate([fA, fB], axis=-1)
    h1 = np.maximum(IO._bn(f @ w["enc_dense_kernel"] + w["enc_dense_bias"], w["enc_bn_gamma"], w["enc_bn_beta"],
                           w["enc_bn_mean"], w["enc_bn_var"]), 0)
    h2 = np.maximum(IO._bn(h1 @ w["head_dense_kernel"] + w["head_dense_bias"], w["head_bn_gamma"], w["head_bn_beta"],
                           w["head_bn_mean"], w["head_bn_var"]), 0)
    return IO._softmax(h2 @ w["out_dense_kernel"] + w["out_dense_bias"])


# The log-normal through the median and the 99th percentile describes weight seed 42 (every window carries a comparable error).  With the
# second seed most windows' scores saturate and carry almost no error while a few carry all of it: the fitted sigma explodes and the
# "extrapolation" exceeds anything the device measures on the full 10^6 windows (profiles/r04_tails.txt is the authoritative tail).
LOGNORMAL_SIGMA_LIMIT = 1.2
EXTRAPOLATION_NOTE = ("bimodal per-window errors (most windows saturated): the log-normal fit is not a bound here; the measured "
                      "1M-window tails on the device (profiles/r04_tails.txt) are authoritative")


def run(n_windows, seeds, names, per_layer, batch=8):
    bases = synthetic.synth_windows(0, n_windows)
    tokens = sequence_oracle.tokenize_closed_form(bases)
    layers = ("conv2", "conv3", "wvA", "wvB")
    rows = []
    for seed in seeds:
        W = synthetic.synth_weights(seed)
        truth = np.concatenate([IO.forward(tokens[a:a + batch], W, dtype=np.float64, literal=False)
                                for a in range(0, n_windows, batch)])
        f32 = np.concatenate([IO.forward(tokens[a:a + batch], W, dtype=np.float32, literal=False)
                              for a in range(0, n_windows, batch)])
        print(f"seed {seed}: fp32 oracle vs fp64 oracle {np.abs(f32 - truth).max():.2e} "
              f"(the f32-accumulation floor every scheme sits on)", flush=True)
        rows.append({"seed": seed, "scheme": "fp32 oracle (f32 everywhere)", "cost": None,
                     "max_abs_dscore": float(np.abs(f32 - truth).max())})

        def measure(label, ls, cost):
            t = time.time()
            got = np.concatenate([forward(tokens[a:a + batch], W, ls) for a in range(0, n_windows, batch)])
            d = np.abs(got - truth).max(axis=1)
            err = float(d.max())
            row = {"seed": seed, "scheme": label, "cost": cost, "max_abs_dscore": err, "rms": float(np.sqrt(np.mean((got - truth) ** 2))),
                   "p99": float(np.quantile(d, 0.99)), "p999": float(np.quantile(d, 0.999)), "windows": int(len(d))}
            # tail extrapolation to 10^6 windows: the per-window maxima of these schemes fall off like a half-normal in log space; the
            # fit below is the log-normal through the median and the 99th percentile, read at the 1 - 1e-6 quantile
            if len(d) >= 1000:
                from statistics import NormalDist
                lm, l99 = np.log(np.median(d)), np.log(np.quantile(d, 0.99))
                sig = (l99 - lm) / NormalDist().inv_cdf(0.99)
                row["extrapolated_max_1M"] = float(np.exp(lm + sig * NormalDist().inv_cdf(1 - 1e-6)))
                if sig > LOGNORMAL_SIGMA_LIMIT:
                    row["extrapolation_note"] = EXTRAPOLATION_NOTE
            rows.append(row)
            print(f"seed {seed}: {label:75s} cost {cost:5.2f}  max|dscore| {err:.2e}  rms {row['rms']:.2e}  p99.9 {row['p999']:.2e}"
                  f"{'  1M-extrapolated ' + format(row['extrapolated_max_1M'], '.2e') if 'extrapolated_max_1M' in row else ''}  ({time.time() - t:.0f} s)", flush=True)

        for name in names:
            s = BY_NAME[name]
            measure(s.name, {l: s for l in layers}, s.cost)
        # per-layer mixes: `base` everywhere, one layer (or both w_v) replaced by `alt`
        for base, alt in per_layer:
            b, a = BY_NAME[base], BY_NAME[alt]
            share = {"conv2": 0.427, "conv3": 0.427, "wvA": 0.071, "wvB": 0.071}
            for group in (("conv2",), ("conv3",), ("wvA", "wvB"), ("conv2", "conv3")):
                ls = {l: (a if l in group else b) for l in layers}
                cost = sum(share[l] * ls[l].cost for l in layers) / sum(share.values())
                measure(f"{b.name}, but {'+'.join(group)}: {a.name}", ls, cost)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=48)
    ap.add_argument("--seeds", type=int, nargs="+", default=[42, 43])
    ap.add_argument("--quick", action="store_true", help="only the headline schemes")
    ap.add_argument("--only", nargs="+", default=None, help="substrings selecting schemes (no per-layer mixes)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_precision_study.json"))
    ap.add_argument("--merge", action="store_true", help="keep the rows --out already holds for other schemes (with --only)")
    args = ap.parse_args()
    names = [s.name for s in SCHEMES]
    per_layer = [("fp16x3", "fp16x1"), ("fp16x3", "fp16x2 (hh+lh: x 2 limbs, w 1)"),
                 ("fp16 + e4m3 corrections (x8*wl8 + xl8*w8)", "fp16x3"),
                 ("fp16 + e2m3 (fp6, MX both sides) corrections", "fp16 + e4m3 corrections (x8*wl8 + xl8*w8)")]
    if args.quick:
        names = [n for n in names if n.startswith(("bf16x3", "fp16x3", "fp16 + e4m3 corrections", "fp16 + e2m3"))]
        per_layer = []
    if args.only:
        names = [n for n in names if any(o in n for o in args.only)]
        per_layer = []
    rows = run(args.windows, args.seeds, names, per_layer)
    if args.merge and os.path.exists(args.out):
        new = {(r["seed"], r["scheme"]) for r in rows}
        rows = [r for r in json.load(open(args.out))["rows"] if (r["seed"], r["scheme"]) not in new] + rows
    with open(args.out, "w") as f:
        json.dump({"windows": args.windows, "tolerance": 1e-4,
                   "note": "max |dscore| vs the fp64 oracle; operands rounded like the hardware formats, products "
                           "and accumulation in f64 (the f32-accumulation floor is the 'fp32 oracle' row)",
                   "toomcook_magnitudes": TC_STATS, "rows": rows}, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
