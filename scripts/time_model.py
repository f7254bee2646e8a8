#!/usr/bin/env python
"""Time / energy model of the default kernel's launch, fitted on the A/B builds of scripts/tc_ab2.py
(profiles/r05/batch3/tc_ab2_cycles.txt: launch time AND the per-step cycle count of every build, one box).

The chip is power-managed: a build that needs C shader cycles per step and switches W joules per step (at a reference voltage) runs
at the clock f at which its power meets the board's budget.  With dynamic power ~ (W / C) f^beta (beta - 1 = the voltage's share:
energy per operation rises with the clock) and a clock ceiling f_max,

        f = min(f_max, f0 * ((C / C0) / (W / W0)) ^ (1 / beta)),        T = C / f

Measured per build: T (HIP events) and C (s_memtime phase counters of the instrumented kernel) -> f = C / T, no power sensor needed
(the hwmon sensors of the pool's boxes disagree: one reads 1 315 W under load and 308 W idle with the true clock, others a constant).
  1. beta from the three SLEEP builds (same W, more C).
  2. every single-part ablation then gives its part's share of W:  W_j / W0 = (C_j / C0) * (f0 / f_j) ^ beta.
  3. validation: builds that remove SEVERAL parts at once (HELPNONE, both low limbs zero) against the sum of their parts.
  4. design space: shares scaled by what a design changes (MFMAs, weight bytes, V reads per row; cycles per row), T predicted.

    python scripts/time_model.py [profiles/r05/batch3/tc_ab2_cycles.txt] [--json out.json]"""
import json
import math
import re
import sys

F_MAX = 2.4e9
STEPS, ROUNDS = 63, 64          # steps per window (all-N tails are skipped: the counters divide by 63 too), rounds of workgroups per 16 384-window launch


def parse(path):
    rows = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+([0-9.]+) ms/4096.*cycles/step matrix\s+(\d+) \[([0-9 ]+)\] helper\s+(\d+)", line)
        if not m:
            continue
        name = m.group(1).replace("build_variants/", "").replace("lib_", "").replace(".so", "")
        rows.setdefault(name, []).append((float(m.group(2)), float(m.group(3)), [float(x) for x in m.group(4).split()]))
    out = {}
    for k, v in rows.items():
        out[k] = {"ms": sum(x[0] for x in v) / len(v), "C": sum(x[1] for x in v) / len(v), "n": len(v),
                  "phases": [sum(x[2][i] for x in v) / len(v) for i in range(8)]}
    return out


def clock(r):
    """effective shader clock: cycles per window / seconds per window (a launch of 16 384 windows = 64 rounds of one window per CU)"""
    return r["C"] * STEPS / (r["ms"] * 4e-3 / ROUNDS)


def main():
    path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "profiles/r05/batch3/tc_ab2_cycles.txt"
    R = parse(path)
    d = R["default"]
    f0, C0, T0 = clock(d), d["C"], d["ms"]
    print(f"default: {T0:.3f} ms per 4096 windows (n={d['n']}), {C0:.0f} cycles per 96-row step, effective clock {f0 / 1e9:.3f} GHz")
    # 1. beta from the sleep builds
    inv = []
    for k in ("tc_sleep16", "tc_sleep48", "tc_sleep96"):
        if k in R:
            r = R[k]
            e = math.log(clock(r) / f0) / math.log(r["C"] / C0)
            inv.append(e)
            print(f"  {k:12s} C x {r['C'] / C0:.4f}  T x {r['ms'] / T0:.4f}  clock {clock(r) / 1e9:.3f} GHz  -> 1/beta = {e:.3f}")
    ib = sum(inv) / len(inv)
    beta = 1 / ib
    alpha = 1 - ib
    print(f"  1/beta = {ib:.3f}  (beta = {beta:.2f});  below the clock ceiling  T ~ C^{alpha:.2f} W^{ib:.2f}")

    def w_ratio(r):
        f = clock(r)
        return (r["C"] / C0) * (f0 / f) ** beta, f

    # 2. shares of W
    parts = [("tcabl_NOCONVMMA", "conv2 + conv3 MFMAs (issue + data), 1 536 per CU and step"),
             ("tcabl_NOWEIGHTS", "L2 -> CU weight stream of the conv loops, 1 MB per CU and step"),
             ("tcabl_NOWV", "y @ w_v: 576 MFMAs, 128 KB of weights, 384 KB of LDS reads"),
             ("tcabl_NOPAIRS", "IGLOO pair products (weff 137 KB L2, rows 137 KB LDS, v_fma_mix)"),
             ("tcabl_NOTRANSFORM", "helpers' B^T + limb split arithmetic (loads / stores kept)"),
             ("tcabl_NOVREAD", "LDS reads of V by the matrix waves, 1 MB per CU and step"),
             ("tcabl_NOEPI", "inverse transform's sums"),
             ("tcabl_GATHER_ONE", "two of the three conv1 table rows (x 1.5 = the gather's L2 reads)"),
             ("tcabl_NOGATHER", "x1 constant: the gather AND the data of conv2 / w_v A / pairs A"),
             ("default,GNN_TC_WLO_MASK=0", "weights' low limbs zero (conv): data of 512 MFMAs per CU-step"),
             ("tc_alo0", "activations' low limbs zero: data of 704 MFMAs per CU-step"),
             ("tc_wva_drop", "w_v A without its x_hi * w_lo product (96 MFMAs per CU-step)")]
    shares = {}
    print("\n  build                       T x      C x     clock GHz   W x      share of W")
    for k, what in parts:
        if k not in R:
            continue
        r = R[k]
        wr, f = w_ratio(r)
        clamp = f >= 0.985 * F_MAX
        shares[k] = 1 - wr
        print(f"  {k:27s} {r['ms'] / T0:.4f}   {r['C'] / C0:.4f}   {f / 1e9:.3f}{'*' if clamp else ' '}     {wr:.4f}   {'>= ' if clamp else '   '}{(1 - wr) * 100:5.1f} %   {what}")
    print("  (* at the 2.4 GHz ceiling: not power-limited any more, the share is a lower bound)")
    single = ["tcabl_NOCONVMMA", "tcabl_NOWEIGHTS", "tcabl_NOWV", "tcabl_NOPAIRS", "tcabl_NOTRANSFORM", "tcabl_NOVREAD", "tcabl_NOEPI"]
    tot = sum(shares[k] for k in single if k in shares) + 1.5 * shares.get("tcabl_GATHER_ONE", 0)
    print(f"  sum of the independent parts: {tot * 100:.1f} % of W; unattributed {100 - tot * 100:.1f} %")

    def predict(c_ratio, w_ratio_):
        f = min(F_MAX, f0 * (c_ratio / w_ratio_) ** ib)
        return c_ratio * C0 * STEPS / f / (4e-3 / ROUNDS)        # ms per 4096 windows

    # 3. validation on combined builds
    print("\n  validation (parts summed, measured C):")
    val = []
    for k, comp in (("tc_helpnone", ["tcabl_NOTRANSFORM", "tcabl_NOPAIRS", "tcabl_NOGATHER"]),
                    ("tc_alo0,GNN_TC_WLO_MASK=0", ["tc_alo0", "default,GNN_TC_WLO_MASK=0"]),
                    ("tc_alo5,GNN_TC_WLO_MASK=FFE0", None)):
        if k not in R:
            continue
        r = R[k]
        if comp:
            w = 1 - sum(shares[c] for c in comp)
            t = predict(r["C"] / C0, w)
            val.append((k, t, r["ms"]))
            print(f"  {k:30s} predicted {t:.3f} ms, measured {r['ms']:.3f} ms ({(t / r['ms'] - 1) * 100:+.1f} %)")
        else:
            wr, f = w_ratio(r)
            print(f"  {k:30s} measured {r['ms']:.3f} ms = {r['ms'] / T0:.4f} x: low limbs cut by 5 mantissa bits save {(1 - wr) * 100:.1f} % of W")
    # every single build reproduces itself by construction; the round trip default -> ablation -> default is exact

    # 4. design space
    s = shares
    conv, wts, wv, vread, tr = s["tcabl_NOCONVMMA"], s["tcabl_NOWEIGHTS"], s["tcabl_NOWV"], s["tcabl_NOVREAD"], s["tcabl_NOTRANSFORM"]
    print("\n  design space (W scaled by what the design changes; C from the phase counters / the pricing probe):")
    ph = d["phases"]           # conv2 loop, epi, wait B1, w_v A, conv3 loop, epi, wait B0, w_v B
    conv_c = ph[0] + ph[4]
    designs = []

    def add(name, dW, dC, note):
        t = predict(1 + dC, 1 - dW)
        designs.append((name, t, dW, dC, note))

    add("shipped: F(3,6), 3 f16 products, 32 tiles in flight", 0, 0, "")
    add("F(2,6) (7 points / 2 rows): MFMAs, weights, V reads x 1.3125", -(conv + wts * 8 / 9 + vread) * 0.3125, conv_c * 0.3125 / C0, "more of everything per row")
    add("F(4,6) (9 points / 4 rows), two row buffers", (conv + wts * 8 / 9 + vread) * (1 - 27 / 32), 0, "does NOT fit the LDS (194.6 KB)")
    # What the single buffer serialises was MEASURED with a timing emulation of its schedule on the shipped kernel (-DTC_EMU_ONEBUF:
    # bit-identical results; head A's pair products before b_7, the helpers' V3 chunks 0, 1 only behind the matrix waves' w_v A, two
    # extra barriers with the x1 store between them; profiles/r05/batch4/ab_emu_onebuf.txt): C 37 862 -> 41 333 (+9.2 %), T 20.92 ->
    # 21.66 ms (+3.5 %; the model: 1.092^0.39 = +3.5 %).  Per 128-row step the same bubbles are amortised over 4/3 the rows: +6.9 %.
    add("F(4,6), ONE row buffer (x1 -> x2 -> x3 in place)", (conv + wts * 8 / 9 + vread) * (1 - 27 / 32), 0.092 * 0.75 + 2500 * 0.75 / C0,
        "serialisation measured by TC_EMU_ONEBUF (+9.2 % cycles per 96-row step) + the conv1 gather exposed between the end-of-step barriers "
        "(~2.5 k cycles: its results have nowhere to wait - helpers 256 / 256 VGPRs, matrix waves 228); conv phases by the probe")
    add("64 tiles in flight (two MFMA blocks per weight fragment)", wts * 8 / 9 * 0.5, 0, "needs 256 accumulator registers per matrix wave: does not fit")
    add("f16 hi*hi + int8 cross terms (2.0 pass eq.)", (conv + wv * 0.6) * (1 / 3) * 0.5, 0, "fails the accuracy gate in emulation; needs a second (i32) accumulator set: does not fit")
    add("low limbs cut by 5 mantissa bits (measured)", 1 - w_ratio(R["tc_alo5,GNN_TC_WLO_MASK=FFE0"])[0] if "tc_alo5,GNN_TC_WLO_MASK=FFE0" in R else 0, 0, "4.9e-5 on 600 windows: no margin")
    add("w_v A on one weight limb (measured)", s.get("tc_wva_drop", 0), R["tc_wva_drop"]["C"] / C0 - 1 if "tc_wva_drop" in R else 0, "1.0e-4 on 600 windows: fails")
    add("persistent grid (measured: launch_tail.txt)", 0, -0.0004, "a round of workgroups alone = 1.004 x the marginal round")
    add("weff in a 24-bit format", s["tcabl_NOPAIRS"] * 0.45 * 0.25, 0, "+2 VALU per weight beside the MFMA stream: negative after the conversion cost")
    for name, t, dW, dC, note in designs:
        print(f"  {name:58s} W {(-dW) * 100:+6.1f} %  C {dC * 100:+5.1f} %  ->  {t:.3f} ms ({(T0 / t - 1) * 100:+5.1f} % windows/s)   {note}")
    if "--json" in sys.argv:
        json.dump({"default": d, "f0_ghz": f0 / 1e9, "inv_beta": ib, "alpha": alpha, "shares_of_W": shares, "validation": val,
                   "designs": [{"name": n, "ms": t, "dW": dW, "dC": dC, "note": note} for n, t, dW, dC, note in designs]},
                  open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
