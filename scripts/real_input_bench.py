"""End-to-end timing of nn_classification.main() on a real (synthetic-content) FASTA file:
write an N-Mbp metagenome-like FASTA, run the drop-in entry point, print per-stage wall times.
Usage: real_input_bench.py [mbp=300]"""
import os, sys, tempfile, time
from pathlib import Path
import numpy as np
sys.path.insert(0, '.')
from genomad_amd import nn_classification as nnc, sequence, synthetic, weights as W

mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tmp = Path(tempfile.mkdtemp(prefix="gnn_real_"))
nwin = mbp * 1_000_000 // 6000
b = synthetic.synth_windows(0, nwin).reshape(-1)
off = synthetic.synth_metagenome_offsets(len(b), seed=5)
fa = tmp / "meta.fna"
with open(fa, "wb") as f:
    for i in range(len(off) - 1):
        f.write(b">contig_%d len=%d\n" % (i, off[i + 1] - off[i]))
        s = b[off[i]:off[i + 1]].tobytes()
        f.write(b"\n".join(s[j:j + 80] for j in range(0, len(s), 80)))
        f.write(b"\n")
size = fa.stat().st_size
wpath = tmp / "weights.npz"
W.save_npz(wpath, synthetic.synth_weights())
os.environ["GENOMAD_AMD_WEIGHTS"] = str(wpath)
print(f"FASTA {size / 1e6:.0f} MB, {len(off) - 1} contigs")
for name, fn in (("check_fasta", lambda: sequence.check_fasta(fa)), ("md5", lambda: nnc.get_md5(fa)),
                 ("read_fasta_packed", lambda: sequence.read_fasta_packed(fa))):
    t = time.time(); fn(); dt = time.time() - t
    print(f"  {name}: {dt:.2f} s = {size / dt / 1e6:.0f} MB/s")
for run in range(2):
    out = tmp / f"out{run}"
    t = time.time()
    nnc.main(fa, out, False, 128, True, 1, False, False)
    dt = time.time() - t
    z = np.load(out / "meta_nn_classification" / "meta_nn_classification.npz")
    print(f"  main() run {run}: {dt:.2f} s = {size / dt / 1e6:.0f} MB/s of FASTA, {len(z['contig_names'])} contigs scored"
          f" (first run includes library/engine start-up and weight upload)")

# where the time goes inside main(): wrap the building blocks with timers and run once more
import collections, functools
from genomad_amd import engine as E
acc = collections.OrderedDict()
def timed(obj, name, label=None):
    f = getattr(obj, name)
    @functools.wraps(f)
    def g(*a, **k):
        t = time.time()
        try:
            return f(*a, **k)
        finally:
            acc[label or name] = acc.get(label or name, 0.0) + time.time() - t
    setattr(obj, name, g)
timed(sequence, "check_fasta"); timed(sequence, "read_fasta_packed"); timed(sequence, "candidate_spans")
timed(E.DeviceBuffer, "upload", "upload packed contigs (H2D)")
timed(E.NNEngine, "segment_mean")
timed(nnc, "write_tsv"); timed(np, "savez_compressed")
eng = nnc._engine()
for name in ("gnn_classify_contigs",):
    timed(eng.lib, name)
t = time.time()
nnc.main(fa, tmp / "out_t", False, 128, True, 1, False, False)
total = time.time() - t
print(f"  breakdown of one main() call ({total:.2f} s):", {k: round(v, 3) for k, v in acc.items()},
      "unaccounted", round(total - sum(v for k, v in acc.items() if k != 'candidate_spans'), 3))
