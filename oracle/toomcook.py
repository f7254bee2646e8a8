"""Toom-Cook minimal filtering F(m, 6) over the time axis for conv2 / conv3 (igloo.py:65-67): exact transform matrices and a
numpy emulation of the arithmetic a split-f16 MFMA kernel would run.

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.  (priced in DESIGN.md section 8.)

The causal 6-tap convolution y[t] = sum_k g[k] x[t + k - 5] is cut into tiles of m outputs.  With n = m + 5 interpolation
points p_0 .. p_{n-2}, infinity:

    y_tile = A^T [ (G g) * (B^T d) ]          d = the tile's n input rows, * = element-wise, per (in, out) channel pair

so the channel contraction becomes n GEMMs of K = 128 per tile instead of 6 m: n / (6 m) of the MFMAs (F(2,6) 0.583, F(3,6)
0.444, F(4,6) 0.375).  A^T[i][j] = p_j^i, G[j][k] = p_j^k (infinity: the last unit vector), B^T = the transposed inverse of the
n x n evaluation matrix (the Lagrange basis polynomials); all three are built in exact rational arithmetic and checked against
the definition of the convolution.  A diagonal scaling D (G <- D G, B^T <- D^-1 B^T) is free; `matrices` picks the one that makes
B^T's rows the monic products prod_{l != j}(x - p_l) (Lavin's form: small dyadic constants, exact in f16 and f32).
"""
from fractions import Fraction as Fr

import numpy as np

POINTS = {2: [0, 1, -1, Fr(1, 2), Fr(-1, 2), 2],
          3: [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2)],
          4: [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2), Fr(1, 4)]}
R = 6


def _inv(mat):
    n = len(mat)
    a = [[Fr(x) for x in row] + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(mat)]
    for c in range(n):
        p = next(r for r in range(c, n) if a[r][c] != 0)
        a[c], a[p] = a[p], a[c]
        pv = a[c][c]
        a[c] = [x / pv for x in a[c]]
        for r in range(n):
            if r != c and a[r][c] != 0:
                f = a[r][c]
                a[r] = [x - f * y for x, y in zip(a[r], a[c])]
    return [row[n:] for row in a]


def matrices(m: int, points=None):
    """Exact (A^T (m x n), G (n x 6), B^T (n x n)) as lists of Fractions, verified against y_i = sum_k g_k d_{i+k}."""
    pts = [Fr(p) for p in (points if points is not None else POINTS[m])]
    n = m + R - 1
    assert len(pts) == n - 1 and len(set(pts)) == n - 1
    ev = lambda deg: [[p ** e for e in range(deg)] for p in pts] + [[Fr(int(e == deg - 1)) for e in range(deg)]]  # noqa: E731
    at = [list(col) for col in zip(*ev(m))]            # A^T = E_m^T
    g = ev(R)
    bt = [list(col) for col in zip(*_inv(ev(n)))]      # B^T = (E_n^-1)^T: row j = coefficients of the Lagrange polynomial of p_j
    # scaling: rows of B^T become monic products (leading non-zero coefficient 1); G absorbs the inverse
    for j in range(n):
        lead = next(x for x in reversed(bt[j]) if x != 0)
        bt[j] = [x / lead for x in bt[j]]
        g[j] = [x * lead for x in g[j]]
    for i in range(m):
        for k in range(R):
            for j in range(n):
                s = sum(at[i][x] * g[x][k] * bt[x][j] for x in range(n))
                assert s == (1 if j == i + k else 0), (i, k, j, s)
    return at, g, bt


def as_float(mat):
    return np.array([[float(x) for x in row] for row in mat], dtype=np.float64)


# ------------------------------------------------------------------ split-f16 emulation
def _f16(x):
    return np.asarray(x, np.float64).astype(np.float16).astype(np.float64)


def _f32(x):
    return np.asarray(x, np.float64).astype(np.float32).astype(np.float64)


def split16(x):
    hi = _f16(x)
    return hi, _f16(np.asarray(x, np.float64) - hi)


def pack_weights(kernel, m: int, points=None, prescale=True):
    """conv kernel (6, C, N) f32 -> transformed weights U (n, C, N) in f64 -> (U_hi, U_lo) f16 limbs, per-point power-of-two
    pre-scales s (U is stored as s * U so that its low limbs leave the f16 subnormal range where they can; A^T absorbs 1/s),
    and max |U| per point."""
    at, g, bt = matrices(m, points)
    G = as_float(g)
    U = np.einsum("xk,kcn->xcn", G, np.asarray(kernel, np.float64))
    n = U.shape[0]
    s = np.ones(n)
    if prescale:
        for x in range(n):
            amax = np.abs(U[x]).max()
            s[x] = 2.0 ** np.floor(np.log2(1024.0 / amax)) if amax > 0 else 1.0      # largest |s U| in [512, 1024): far inside f16
    Us = U * s[:, None, None]
    hi, lo = split16(Us)
    return {"at": as_float(at) / s[None, :], "bt": as_float(bt), "U_hi": hi, "U_lo": lo, "scale": s,
            "max_U": np.abs(U).max(axis=(1, 2)), "m": m, "n": n}


def mask16(lo, bits):
    """f16 low limb with its `bits` lowest mantissa bits cleared (what `lo & mask` does to the f16 word: truncation towards zero)."""
    if not bits:
        return lo
    h = np.asarray(lo, np.float64).astype(np.float16)
    return (h.view(np.uint16) & np.uint16((0xFFFF << bits) & 0xFFFF)).view(np.float16).astype(np.float64)


def conv_emulated(x, pk, bias, kunit=16, stats=None, lo_mask_bits=0):
    """Causal conv of x (B, T, C) (f32-representable values) with packed weights `pk`, emulating the kernel:

      * B^T d in f32 (every operation rounded to f32), V split into f16 hi / lo;
      * three MFMA products per k16 unit (U_lo V_hi, U_hi V_hi, U_hi V_lo), each 16-term dot product exact and the running sum
        rounded to f32 after every MFMA (the f32 accumulator);
      * A^T M in f32, + bias.
    Returns (B, T, N) in f64 holding f32 values.  `stats` (dict) collects max |V|, max |M|."""
    m, n = pk["m"], pk["n"]
    B, T, C = x.shape
    nt = -(-T // m)
    xp = np.zeros((B, (nt - 1) * m + n, C))
    xp[:, R - 1:R - 1 + T] = x
    d = [xp[:, j:j + (nt - 1) * m + 1:m] for j in range(n)]          # d[j][:, t] = input row m t + j - 5
    bt = pk["bt"]
    V = []
    for xi in range(n):
        acc = None
        for j in range(n):
            c = bt[xi, j]
            if c == 0:
                continue
            term = _f32(c * d[j])                                    # products by dyadic constants are exact or one rounding
            acc = term if acc is None else _f32(acc + term)
        V.append(acc)
    V = np.stack(V)                                                  # (n, B, nt, C)
    Vh, Vl = split16(V)
    Vl, Ulo = mask16(Vl, lo_mask_bits), mask16(pk["U_lo"], lo_mask_bits)       # round 5 energy probe: shorter low limbs
    if stats is not None:
        stats["max_V"] = max(stats.get("max_V", 0.0), float(np.abs(V).max()))
    Vh, Vl = Vh.reshape(n, B * nt, C), Vl.reshape(n, B * nt, C)
    N = pk["U_hi"].shape[2]
    M = np.zeros((n, B * nt, N))
    for u in range(0, C, kunit):
        sl = slice(u, u + kunit)
        for a, w in ((Vh, Ulo), (Vh, pk["U_hi"]), (Vl, pk["U_hi"])):
            M = _f32(M + np.matmul(a[:, :, sl], w[:, sl, :]))
    if stats is not None:
        stats["max_M"] = max(stats.get("max_M", 0.0), float(np.abs(M).max()))
    at = pk["at"]
    y = np.zeros((B, nt * m, N))
    Mr = M.reshape(n, B, nt, N)
    for i in range(m):
        acc = None
        for xi in range(n):
            c = at[i, xi]
            if c == 0:
                continue
            term = _f32(c * Mr[xi])
            acc = term if acc is None else _f32(acc + term)
        y[:, i::m] = acc
    return _f32(y[:, :T] + bias)


def conv_direct_x3(x, kernel, bias, kunit=16):
    """The arithmetic of today's kernel (gnn_fused_x3.hip) under the same emulation: direct 6-tap conv, operands split into f16
    limbs, three products per k16 unit, f32 accumulator rounded after every MFMA."""
    B, T, C = x.shape
    K, _, N = kernel.shape
    wh, wl = split16(np.asarray(kernel, np.float64))
    xh, xl = split16(x)
    pad = lambda a: np.concatenate([np.zeros((B, K - 1, C)), a], axis=1)     # noqa: E731
    xh, xl = pad(xh), pad(xl)
    acc = np.broadcast_to(np.asarray(bias, np.float64), (B, T, N)).copy()
    for k in range(K):
        for u in range(0, C, kunit):
            sl = slice(u, u + kunit)
            for a, w in ((xh, wl), (xh, wh), (xl, wh)):
                acc = _f32(acc + np.matmul(a[:, k:k + T, sl], w[k, sl, :]))
    return acc


# ------------------------------------------------------------------ f16 hi*hi + int8 cross terms (emulation only)
def q8_rows(v, axis, tied_to=None):
    """Block-scaled int8 image of v: integers in [-127, 127] times a scale shared along `axis` (the MFMA's K axis: an i32
    accumulator cannot mix scales inside one dot product).  tied_to = the scale array of the operand's HIGH part: the image of a
    residual then uses that scale * 2^-11, so that x8 * wl8 and xl8 * w8 share ONE product scale and ONE i32 accumulator."""
    v = np.asarray(v, np.float64)
    if tied_to is None:
        amax = np.abs(v).max(axis=axis, keepdims=True)
        scale = np.where(amax > 0, amax, 1.0) / 127.0
    else:
        scale = tied_to * 2.0 ** -11
    q = np.clip(np.rint(v / scale), -127, 127)
    return q, scale


def conv_emulated_i8cross(x, pk, bias, kunit=16, stats=None):
    """conv_emulated with the two CROSS products (V_hi U_lo + V_lo U_hi) replaced by int8 MFMAs: v_mfma_i32_32x32x32_i8 runs at
    twice the f16 rate, so the arithmetic costs 1 + 2 x 0.5 = 2.0 pass equivalents instead of 3.0.  V is quantised per (point,
    tile) over its 128 channels, U per (point, output channel) over its 128 input channels; the residual images use the tied
    scales of q8_rows, the integer dot products over K = 128 are exact (|sum| < 2^22), one f32 multiply by the product scale and one
    f32 add bring them into the f32 accumulator of the hi*hi MFMAs."""
    m, n = pk["m"], pk["n"]
    B, T, C = x.shape
    nt = -(-T // m)
    xp = np.zeros((B, (nt - 1) * m + n, C))
    xp[:, R - 1:R - 1 + T] = x
    d = [xp[:, j:j + (nt - 1) * m + 1:m] for j in range(n)]
    bt = pk["bt"]
    V = []
    for xi in range(n):
        acc = None
        for j in range(n):
            c = bt[xi, j]
            if c == 0:
                continue
            term = _f32(c * d[j])
            acc = term if acc is None else _f32(acc + term)
        V.append(acc)
    V = np.stack(V).reshape(n, B * nt, C)
    Vh = _f16(V)
    Us = pk["U_hi"] + pk["U_lo"]                  # s U to 22 bits: what the packer holds before it chooses the limb formats
    Uh = pk["U_hi"]
    qV, aV = q8_rows(V, axis=2)                   # (n, tiles, C), scale (n, tiles, 1)
    qVl, _ = q8_rows(V - Vh, axis=2, tied_to=aV)
    qU, bU = q8_rows(Us, axis=1)                  # (n, C, N), scale (n, 1, N)
    qUl, _ = q8_rows(Us - Uh, axis=1, tied_to=bU)
    N = Uh.shape[2]
    M = np.zeros((n, B * nt, N))
    for u in range(0, C, kunit):
        sl = slice(u, u + kunit)
        M = _f32(M + np.matmul(Vh[:, :, sl], Uh[:, sl, :]))
    cross = np.matmul(qV, qUl) + np.matmul(qVl, qU)                       # exact integers
    M = _f32(M + _f32(cross * _f32(aV * bU * 2.0 ** -11)))
    if stats is not None:
        stats["max_abs_i32"] = max(stats.get("max_abs_i32", 0.0), float(np.abs(cross).max()))
    at = pk["at"]
    y = np.zeros((B, nt * m, N))
    Mr = M.reshape(n, B, nt, N)
    for i in range(m):
        acc = None
        for xi in range(n):
            c = at[i, xi]
            if c == 0:
                continue
            term = _f32(c * Mr[xi])
            acc = term if acc is None else _f32(acc + term)
        y[:, i::m] = acc
    return _f32(y[:, :T] + bias)
