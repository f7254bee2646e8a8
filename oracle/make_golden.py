"""Generate tests/golden/* (run in the build container, where /root/reference exists).

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.

    python -m oracle.make_golden

* tokenizer_golden.json, windowing_golden.json, fasta_golden.json (+ fasta_fixture.fna.gz):
  outputs of the REFERENCE's own functions (genomad/sequence.py executed in place through
  oracle/reference_harness.py): tokenize_dna, seq_windows, read_fasta, Sequence.count /
  .seq_ascii, chained exactly as generate_data does (modules/nn_classification.py:54-82).
* forward_golden.npz: for the first 16 synthetic windows with the seed-42 synthetic weights,
  - ``scores_refgraph64`` / ``scores_refgraph32``: outputs of the REFERENCE'S OWN network code —
    genomad/neural_network/model.py create_classifier() and igloo.py executed in place through
    reference_harness.reference_classifier_scores(), TensorFlow/Keras primitives supplied by the
    numpy stand-ins of oracle/keras_shim.py (TensorFlow itself is not installable here);
  - ``scores`` and the intermediates: the fp64 oracle's own outputs (regression anchors for the
    taps the reference graph does not expose).
"""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from genomad_amd import synthetic  # noqa: E402
from oracle import igloo_oracle, reference_harness, sequence_oracle  # noqa: E402


def tokenizer_cases(rng):
    cases = [b"ACGTACGTNACGTAC", b"NACGT", b"ACGTN", b"AAAAC", b"TTTT", b"ACG", b"", b"A",
             b"NNNN", b"ACGTNNNNACGT", b"acgtACGT", b"ACGURYKMACGT", b"ACGT-ACGT", b"NACGTACGTN",
             b"ANCNGNTNACGTACGT", (b"ACGT" * 700).ljust(6000, b"N")[2790:2810]]
    alphabets = [b"ACGT", b"ACGTN", b"ACGTNRYKMSWacgtn-", bytes(range(256))]
    for k in range(60):
        alpha = alphabets[k % len(alphabets)]
        n = int(rng.integers(1, 80))
        cases.append(bytes(alpha[i] for i in rng.integers(0, len(alpha), n)))
    # mostly-ACGT strings with sparse non-ACGT bytes at the ends and inside
    for k in range(20):
        n = int(rng.integers(8, 120))
        s = bytearray(b"ACGT"[i] for i in rng.integers(0, 4, n))
        for pos in rng.integers(0, n, int(rng.integers(0, 4))):
            s[pos] = ord("N")
        if k % 4 == 0:
            s[0] = ord("N")
        if k % 4 == 1:
            s[-1] = ord("N")
        cases.append(bytes(s))
    return cases


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    ref = reference_harness.load_reference_sequence()
    rng = np.random.default_rng(20260925)

    tok = [{"seq_hex": c.hex(), "tokens": [int(t) for t in ref.tokenize_dna(c, 4)]}
           for c in tokenizer_cases(rng)]
    # full-size windows: store only a digest of the 5997 tokens plus the first/last few
    full = []
    for i in (0, 5, 9, 21):
        w = bytes(synthetic.synth_windows(i, 1)[0])
        t = np.array(ref.tokenize_dna(w, 4), dtype=np.uint16)
        full.append({"synthetic_index": i, "n_tokens": int(len(t)),
                     "sha256_u16le": hashlib.sha256(t.astype("<u2").tobytes()).hexdigest(),
                     "head": [int(x) for x in t[:8]], "tail": [int(x) for x in t[-8:]],
                     "n_zero": int((t == 0).sum())})
    with open(os.path.join(GOLDEN, "tokenizer_golden.json"), "w") as f:
        json.dump({"source": "genomad/sequence.py:170-193 tokenize_dna(seq, 4), executed in place",
                   "cases": tok, "synthetic_windows": full}, f, indent=0)

    lens = [1, 2, 100, 2499, 2500, 2501, 5999, 6000, 6001, 8499, 8500, 8501, 12000, 14499, 14500,
            15000, 18000, 20499, 20500]
    wins = {}
    for L in lens:
        s = ref.Sequence("x", "A" * L)
        wins[str(L)] = {"all": [len(w) for w in ref.seq_windows(s, 6000, 2500)],
                        "single": [len(w) for w in ref.seq_windows(s, 6000, 2500, max_windows=1)]}
    with open(os.path.join(GOLDEN, "windowing_golden.json"), "w") as f:
        json.dump({"source": "genomad/sequence.py:150-167 seq_windows(seq, 6000, 2500[, max_windows=1])",
                   "window_lengths": wins}, f, indent=0)

    # FASTA fixture exercising: multi-line records, lower case, terminal N stripping, an all-N
    # record (dropped), internal N-rich windows (skip rule), short contigs, text before the first '>'.
    def rnd(n, alpha="ACGT"):
        return "".join(alpha[i] for i in rng.integers(0, len(alpha), n))
    recs = [
        ("contigA desc one", rnd(14500)),
        ("contigB", "NNnn" + rnd(7000) + "nnNN"),
        ("allN", "N" * 300),
        ("contigC lower", rnd(6100, "acgt")),
        ("contigD nrich", rnd(6000) + "N" * 4500 + rnd(1500) + rnd(3000)),
        ("contigE lowern", rnd(6000) + "n" * 4500 + rnd(1500)),
        ("tiny", "ACG"),
        ("contigF iupac", rnd(3000, "ACGTRYKM")),
        ("contigG", rnd(12000 + 2499)),
    ]
    fasta = "; a comment line before the first record\n"
    for name, seq in recs:
        fasta += f">{name}\n"
        width = 70
        fasta += "\n".join(seq[i:i + width] for i in range(0, len(seq), width)) + "\n"
    fpath = os.path.join(GOLDEN, "fasta_fixture.fna.gz")
    with gzip.GzipFile(fpath, "wb", mtime=0) as f:
        f.write(fasta.encode())

    out = {}
    for single in (False, True):
        names, ids, digests, ncount = [], [], [], []
        max_windows = 1 if single else None
        # modules/nn_classification.py:65-76, calling the reference's own functions
        for contig_id, seq in enumerate(ref.read_fasta(fpath, strip_n=True)):
            names.append(seq.accession)
            for window_n, sw in enumerate(ref.seq_windows(seq, 6000, 2500, max_windows=max_windows)):
                if window_n > 0 and sw.count("N") > 4000:
                    continue
                padded = sw.seq_ascii.ljust(6000, b"N")
                t = np.array(ref.tokenize_dna(padded, 4), dtype="<u2")
                ids.append(contig_id)
                digests.append(hashlib.sha256(padded).hexdigest()[:16] + ":" + hashlib.sha256(t.tobytes()).hexdigest()[:16])
                ncount.append(int(padded.count(b"N")))
        out["single" if single else "all"] = {"contig_names": names, "contig_ids": ids,
                                              "window_digests": digests, "window_n_count": ncount}
    out["check_fasta"] = bool(ref.check_fasta(fpath))
    with open(os.path.join(GOLDEN, "fasta_golden.json"), "w") as f:
        json.dump({"source": "reference read_fasta/seq_windows/Sequence/tokenize_dna chained as "
                             "modules/nn_classification.py:54-82 on fasta_fixture.fna.gz", **out}, f, indent=0)

    W = synthetic.synth_weights()
    bases = synthetic.synth_windows(0, 16)
    tokens = sequence_oracle.tokenize_closed_form(bases)
    scores, taps = igloo_oracle.forward(tokens, W, dtype=np.float64, return_taps=True)
    ref64 = reference_harness.reference_classifier_scores(tokens, W, np.float64)
    ref32 = reference_harness.reference_classifier_scores(tokens, W, np.float32)
    print("oracle fp64 vs reference graph fp64: max |d| = %.3e" % np.abs(scores - ref64).max())
    np.savez_compressed(
        os.path.join(GOLDEN, "forward_golden.npz"),
        scores_refgraph64=ref64, scores_refgraph32=ref32,
        scores=scores, feat=taps["f"], logits=taps["logits"],
        mA=taps["mA"], mB=taps["mB"], alphaA=taps["alphaA"], alphaB=taps["alphaB"],
        x1_rows=taps["x1"][:, [0, 1, 5, 2999, 5996]], x3_rows=taps["x3"][:, [0, 1, 5, 2999, 5996]],
        ypA_rows=taps["ypA"][:, [0, 374, 748]], ypB_rows=taps["ypB"][:, [0, 374, 748]],
        bases_sha256=np.array(hashlib.sha256(bases.tobytes()).hexdigest()),
        weights_sha256=np.array(hashlib.sha256(b"".join(W[k].tobytes() for k in sorted(W))).hexdigest()))
    # Downstream consumers: plain numpy in the reference, so the REFERENCE'S OWN
    # functions run here and pin the device epilogue.
    import importlib
    agg = importlib.import_module("genomad.modules.aggregated_classification")
    cal = importlib.import_module("genomad.modules.score_calibration")
    wfile = os.path.join(reference_harness.REFERENCE_ROOT, "genomad", "data", "score_calibration_weights.npz")
    n = 257
    w = rng.random(n) * 0.6
    w[:3] = [0.0, 1.0, 0.5]
    b1 = rng.dirichlet([1, 1, 1], n)
    b2 = rng.dirichlet([0.3, 0.3, 0.3], n)
    fixture = {"w": w, "b1": b1, "b2": b2,
               "branch_attention_t2": agg.branch_attention(w, b1, b2),
               "branch_attention_t1": agg.branch_attention(w, b1, b2, temperature=1)}
    comps = np.array([[0.7, 0.1, 0.2], [1 / 3, 1 / 3, 1 / 3], [0.98, 0.01, 0.01], [0.0, 0.0, 1.0]])
    fixture["compositions"] = comps
    for ci, comp in enumerate(comps):
        for classifier in ("nn", "marker", "aggregated", "something_else"):
            fixture[f"calibrated_{ci}_{classifier}"] = cal.score_batch_correction(b2, comp, classifier, wfile)
    with np.load(wfile) as z:                      # the MLP weights are data the test needs on the GPU box
        for k in z.files:
            fixture[f"weights__{k}"] = z[k]
    np.savez_compressed(os.path.join(GOLDEN, "consumers_golden.npz"), **fixture)
    print("golden fixtures written to", GOLDEN)


if __name__ == "__main__":
    main()
