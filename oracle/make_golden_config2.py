"""Generate tests/golden/config2_golden.npz: BASELINE config 2 pinned on all 10 000 windows.

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.  Run in the build container (needs /root/reference):

    python -m oracle.make_golden_config2 [--n 10000] [--chunk 50] [--threads 4]
    python -m oracle.make_golden_config2 --n 16384 --stride 64 --chunk 64 --out tests/golden/config3_strided_golden.npz \
        --scratch /tmp/config3_golden        # BASELINE config 3: every 64th of its 1 048 576 windows (round 5; 80 CPU-minutes on
                                             # 4 threads; rounds 3-4 held every 512th: --n 2048 --stride 512)

For synthetic windows 0 … n-1 (seed 1234) and the seed-42 synthetic weights it stores

* ``scores_refgraph32`` (n, 3) f32 — outputs of the REFERENCE'S OWN graph: genomad/neural_network/
  model.py:34-45 ``create_classifier()`` + igloo.py executed in place in float32 (as the reference
  runs) over the numpy stand-ins of oracle/keras_shim.py, ``predict`` called per chunk like
  modules/nn_classification.py:316-318;
* ``scores_oracle64`` (n, 3) f64 — the fp64 restatement (oracle/igloo_oracle.py), i.e. the value both
  float32 evaluations approximate.

The run is resumable: finished chunks are kept under ``--scratch`` (default /tmp/config2_golden) and
only merged into the fixture at the end.  ≈ 0.3 s per window on 4 threads.
"""
import argparse
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--chunk", type=int, default=50)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--stride", type=int, default=1, help="window j of the fixture is synthetic window first + j * stride")
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--scratch", default="/tmp/config2_golden")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "config2_golden.npz"))
    args = ap.parse_args()
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = str(args.threads)
    import numpy as np
    from genomad_amd import synthetic
    from oracle import igloo_oracle, reference_harness, sequence_oracle

    os.makedirs(args.scratch, exist_ok=True)
    W = synthetic.synth_weights()
    r32, o64 = [], []
    for a in range(0, args.n, args.chunk):
        b = min(a + args.chunk, args.n)
        path = os.path.join(args.scratch, f"chunk_{a:06d}_{b:06d}.npz")
        if os.path.exists(path):
            with np.load(path) as z:
                r32.append(z["r32"]), o64.append(z["o64"])
            continue
        if args.stride == 1:
            bases = synthetic.synth_windows(args.first + a, b - a)
        else:
            bases = np.concatenate([synthetic.synth_windows(args.first + j * args.stride, 1) for j in range(a, b)])
        tokens = sequence_oracle.tokenize_closed_form(bases)
        ref = reference_harness.reference_classifier_scores(tokens, W, np.float32).astype(np.float32)
        orc = igloo_oracle.forward(tokens, W, dtype=np.float64, literal=False)
        np.savez(path + ".tmp.npz", r32=ref, o64=orc)
        os.replace(path + ".tmp.npz", path)
        r32.append(ref), o64.append(orc)
        print(f"{b}/{args.n}  max|ref32-oracle64| so far {max(np.abs(x - y).max() for x, y in zip(r32, o64)):.3e}",
              flush=True)
    r32, o64 = np.concatenate(r32), np.concatenate(o64)
    np.savez_compressed(
        args.out, scores_refgraph32=r32, scores_oracle64=o64, n=np.array(args.n),
        indices=args.first + np.arange(args.n, dtype=np.int64) * args.stride,
        data_seed=np.array(synthetic.DATA_SEED if hasattr(synthetic, "DATA_SEED") else 1234),
        weights_sha256=np.array(hashlib.sha256(b"".join(W[k].tobytes() for k in sorted(W))).hexdigest()))
    print("wrote", args.out, r32.shape, "max|ref32-oracle64| = %.3e" % np.abs(r32 - o64).max())


if __name__ == "__main__":
    main()
