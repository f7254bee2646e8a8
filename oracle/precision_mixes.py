"""Per-layer mixes of MFMA operand schemes: is there an arithmetic cheaper than three f16 passes that keeps the class scores
inside 1e-4 WITH MARGIN?  

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.  numpy emulation on top of oracle/precision_study.py (same operand rounding,
same forward pass; products and sums in f64), no GPU:

    python -m oracle.precision_mixes [windows, default 64] > profiles/history/r03_precision_mixes.txt

Each of the four matrix-pipe contractions (conv2, conv3, y @ w_v of head A / head B) gets its own scheme; the statistic is the
rms of the score error over windows x classes next to the maximum (the maximum of a few dozen windows moves by 2x between
seeds and cannot rank schemes that differ by tens of percent).  Cost = MFMA passes weighted by the layers' FLOP shares.
"""
import sys
from multiprocessing import Pool

import numpy as np

from genomad_amd import synthetic
from oracle import igloo_oracle as IO
from oracle import precision_study as PS
from oracle import sequence_oracle

c6 = PS.BY_NAME["fp16 + e2m3 (fp6, MX both sides) corrections"]
x3 = PS.BY_NAME["fp16x3"]
p25 = PS.BY_NAME["fp16 x (w 2 limbs) + e4m3 correction of x (xh*wh + xh*wl16 + xl8*w8)"]
MIXES = {
    "c6 / c6 / c6 / c6  (f16c6)":              dict(conv2=c6, conv3=c6, wvA=c6, wvB=c6),
    "c6 / c6 / c6 / x3":                       dict(conv2=c6, conv3=c6, wvA=c6, wvB=x3),
    "x3 / c6 / c6 / c6":                       dict(conv2=x3, conv3=c6, wvA=c6, wvB=c6),
    "c6 / x3 / c6 / x3":                       dict(conv2=c6, conv3=x3, wvA=c6, wvB=x3),
    "2.5-pass convs and w_v B, c6 w_v A":      dict(conv2=p25, conv3=p25, wvA=c6, wvB=p25),
    "x3 / x3 / c6 / c6":                       dict(conv2=x3, conv3=x3, wvA=c6, wvB=c6),
    "x3 / x3 / c6 / x3":                       dict(conv2=x3, conv3=x3, wvA=c6, wvB=x3),
    "x3 / x3 / x3 / x3  (f16x3)":              dict(conv2=x3, conv3=x3, wvA=x3, wvB=x3),
}
SHARE = {"conv2": 0.427, "conv3": 0.427, "wvA": 0.071, "wvB": 0.071}
_STATE = {}


def _init(n):
    _STATE["tokens"] = sequence_oracle.tokenize_closed_form(synthetic.synth_windows(0, n))
    _STATE["W"] = synthetic.synth_weights(42)


def _work(args):
    name, a = args
    tok, W = _STATE["tokens"][a:a + 8], _STATE["W"]
    if name == "truth":
        return name, a, IO.forward(tok, W, dtype=np.float64, literal=False)
    return name, a, PS.forward(tok, W, MIXES[name])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    jobs = [(m, a) for m in ["truth"] + list(MIXES) for a in range(0, n, 8)]
    with Pool(8, initializer=_init, initargs=(n,)) as pool:
        res = pool.map(_work, jobs)
    out = {}
    for m, a, s in res:
        out.setdefault(m, {})[a] = s
    cat = {m: np.concatenate([out[m][a] for a in sorted(out[m])]) for m in out}
    print(f"{n} synthetic windows, weight seed 42; conv2 / conv3 / w_v A / w_v B; error of the class scores against the fp64 oracle")
    for m, layers in MIXES.items():
        e = cat[m] - cat["truth"]
        cost = sum(SHARE[k] * layers[k].cost for k in SHARE) / sum(SHARE.values())
        print(f"{m:40s} passes {cost:4.2f}  rms {np.sqrt((e ** 2).mean()):.2e}  max {np.abs(e).max():.2e}  "
              f"99th pct {np.quantile(np.abs(e), 0.99):.2e}", flush=True)


if __name__ == "__main__":
    main()
