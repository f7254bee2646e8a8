"""Drop-in replacement for ``genomad.modules.nn_classification`` on MI355X.

    genomad_amd.nn_classification.main(input_path, output_path, single_window, batch_size,
                                       restart, threads, verbose, cleanup)

has the signature, the on-disk outputs, the resume rules and the error behaviour of the
reference's ``main`` (genomad/modules/nn_classification.py:21-427); ``install()`` rebinds
``genomad.nn_classification`` / ``genomad.modules.nn_classification`` so that ``genomad
nn-classification`` and ``genomad end-to-end`` (cli.py:772, :1367-1376) pick it up unchanged.

What differs inside: windows are tokenised and classified by libgenomad_nn_hip.so (no TensorFlow,
no TFRecord round trip).  ``<prefix>_encoded_sequences/`` holds ``<prefix>_seq_window_id.npz``
(same keys as the reference) and the padded windows as ``<n>.win.npy`` instead of ``*.tfrec``.
``threads`` is accepted and ignored (it only sized TensorFlow's pools, :39-40); ``batch_size``
bounds the windows per GPU launch (results do not depend on it).

Weights: ``$GENOMAD_AMD_WEIGHTS`` or ``<genomad data dir>/nn_classifier.npz`` in the schema of
genomad_amd/weights.py.  With WORLD_SIZE > 1 (one process per GPU, e.g. started by
``python -m torch.distributed.run`` — used only as a process launcher) the contigs are sharded across
ranks, the per-contig scores are collected with one RCCL gather (genomad_amd/rccl.py, no torch) and rank 0
writes the files.
"""
import concurrent.futures
import hashlib
import json
import os
import shutil
import sys
import threading
import time
from datetime import datetime, timezone
from pathlib import Path

import numpy as np

from . import sequence
from ._lib import DEFAULT_PRECISION

# arithmetic of the fused front end (GENOMAD_AMD_PRECISION overrides): "f16x3tc" = split-f16 limbs with conv2 / conv3 by Toom-Cook
# minimal filtering, f32-class accuracy - the default, because the TSV prints four decimals and the 1e-4 tolerance has to hold
# with margin on inputs and weights nobody has measured; "f16x3" = the direct three-pass form (round 3's default, 0.82x the
# speed); "f16c6" / "f16c8" = f16 MFMA + MX-fp6 / fp8 corrections (faster, no head-room: DESIGN.md section 2); "bf16x3" =
# split-bf16 three passes (f32 range); "f32" = exact f32 reference kernels
MODULE_NAME = "nn_classification"   # utils.write_execution_info("nn_classification", ...) :207-212
TSV_HEADER = "seq_name\tchromosome_score\tplasmid_score\tvirus_score\n"   # :345


class Outputs:
    """File naming contract, genomad/_paths.py:188-236 (+ the find-proviruses files that
    utils.check_provirus_execution reads, utils.py:280-297)."""

    def __init__(self, prefix: str, output_dir: Path):
        o, p = Path(output_dir), prefix
        self.nn_classification_log = o / f"{p}_nn_classification.log"
        d = self.nn_classification_dir = o / f"{p}_nn_classification"
        self.nn_classification_execution_info = d / f"{p}_nn_classification.json"
        self.encoded_sequences_dir = d / f"{p}_encoded_sequences"
        self.seq_window_id_output = self.encoded_sequences_dir / f"{p}_seq_window_id.npz"
        self.nn_classification_output = d / f"{p}_nn_classification.tsv"
        self.nn_classification_npz_output = d / f"{p}_nn_classification.npz"
        self.encoded_proviruses_dir = d / f"{p}_encoded_proviruses"
        self.provirus_window_id_output = self.encoded_proviruses_dir / f"{p}_provirus_window_id.npz"
        self.provirus_nn_classification_output = d / f"{p}_provirus_nn_classification.tsv"
        self.provirus_nn_classification_npz_output = d / f"{p}_provirus_nn_classification.npz"
        f = o / f"{p}_find_proviruses"
        self.find_proviruses_execution_info = f / f"{p}_find_proviruses.json"
        self.find_proviruses_output = f / f"{p}_provirus.tsv"
        self.find_proviruses_nucleotide_output = f / f"{p}_provirus.fna"
        self.find_proviruses_proteins_output = f / f"{p}_provirus_proteins.faa"
        self.find_proviruses_genes_output = f / f"{p}_provirus_genes.tsv"


class Console:
    """Minimal stand-in for utils.HybridConsole (utils.py:42-123): messages go to stdout (unless
    quiet) and are appended to the log file, which is deleted at construction if it exists."""

    def __init__(self, output_file=None, verbose=True):
        self.output_file, self.verbose = output_file, verbose
        if output_file and Path(output_file).exists():
            Path(output_file).unlink()

    def _emit(self, msg, stream):
        line = f"[{datetime.now().strftime('%X')}] {msg}"
        if stream is not None:
            print(line, file=stream, flush=True)
        if self.output_file:
            with open(self.output_file, "a") as fout:
                fout.write(line + "\n")

    def log(self, msg, **_):
        self._emit(msg, sys.stdout if self.verbose else None)

    def error(self, msg, **_):
        self._emit(msg, sys.stderr)


def _md5_of(path, size=1 << 22) -> str:   # utils.py:216-223 (same digest; bigger blocks)
    m = hashlib.md5()
    with open(path, "rb") as fin:
        for block in iter(lambda: fin.read(size), b""):
            m.update(block)
    return m.hexdigest()


_MD5_FUTURES = {}
_MD5_LOCK = threading.Lock()


def md5_async(path) -> "concurrent.futures.Future":
    """The md5 of ``path`` as a future, computed once per (path, size, mtime) on a background thread.
    The reference hashes the input up to three times per run (utils.py:241, :277, :284) at
    ≈ 0.9 GB/s — comparable to the whole GPU classification of the same file — so main() starts it
    first and everything that needs the digest waits on the same future (hashlib releases the GIL)."""
    st = os.stat(path)
    key = (str(Path(path).resolve()), st.st_size, st.st_mtime_ns)
    with _MD5_LOCK:
        fut = _MD5_FUTURES.get(key)
        if fut is None:
            if len(_MD5_FUTURES) > 8:
                _MD5_FUTURES.clear()
            fut = concurrent.futures.Future()
            _MD5_FUTURES[key] = fut

            def work():
                try:
                    fut.set_result(_md5_of(path))
                except BaseException as exc:  # noqa: BLE001
                    fut.set_exception(exc)
            threading.Thread(target=work, name="genomad-amd-md5").start()
    return fut


def get_md5(path, size=None) -> str:      # utils.py:216-223
    return md5_async(path).result()


def write_execution_info(module_name, input_file: Path, parameters: dict, output_file: Path, start_time=None):
    """utils.py:238-254, byte for byte (indent=4, trailing newline, local-tz ISO start time)."""
    start_time = start_time or datetime.now(timezone.utc).astimezone().isoformat()
    dump = json.dumps({"module": module_name, "input": Path(input_file).name,
                       "input_md5": get_md5(input_file),
                       "start_time": start_time,
                       "parameters": parameters}, indent=4)
    tmp = Path(f"{output_file}.tmp{os.getpid()}")      # never leave a half-written JSON for a concurrent reader
    with open(tmp, "w") as fout:
        fout.write(f"{dump}\n")
    os.replace(tmp, output_file)


def compare_executions(input_file, parameters, execution_info_file) -> bool:   # utils.py:266-277
    with open(execution_info_file) as fin:
        info = json.load(fin)
    return parameters == info["parameters"] and get_md5(input_file) == info["input_md5"]


def check_provirus_execution(outputs: Outputs, input_file) -> bool:   # utils.py:280-297
    if not outputs.find_proviruses_execution_info.exists():
        return False
    with open(outputs.find_proviruses_execution_info) as fin:
        if get_md5(input_file) != json.load(fin)["input_md5"]:
            return False
    required = [outputs.find_proviruses_output, outputs.find_proviruses_nucleotide_output,
                outputs.find_proviruses_proteins_output, outputs.find_proviruses_genes_output]
    if not all(p.exists() for p in required):
        return False
    with sequence.open_text(outputs.find_proviruses_output) as fin:
        next(fin, None)
        return sum(1 for _ in fin) > 0


def write_tsv(path, names, predictions):
    """nn_classification.py:344-348: '%.4f' columns joined by tabs, no trailing tab."""
    p = np.asarray(predictions)
    with open(path, "w") as fout:
        fout.write(TSV_HEADER)
        if p.ndim == 2 and p.shape[1] == 3:      # the module's only shape: one formatted write per 64 Ki contigs
            cols = [p[:, j].tolist() for j in range(3)]        # exact float32 -> float conversions: the same digits
            for lo in range(0, len(p), 1 << 16):
                hi = lo + (1 << 16)
                fout.write("".join(f"{n}\t{a:.4f}\t{b:.4f}\t{c:.4f}\n" for n, a, b, c in
                                   zip(names[lo:hi], cols[0][lo:hi], cols[1][lo:hi], cols[2][lo:hi])))
        else:
            for name, scores in zip(names, p):
                row = "\t".join(f"{x:.4f}" for x in scores)
                fout.write(f"{name}\t{row}\n")


def find_weights() -> Path:
    cand = []
    if os.environ.get("GENOMAD_AMD_WEIGHTS"):
        cand.append(Path(os.environ["GENOMAD_AMD_WEIGHTS"]))
    try:
        from genomad._paths import GenomadData   # only when the reference package is installed
        cand.append(Path(GenomadData.nn_model_file).with_suffix(".npz"))
        cand.append(Path(GenomadData.nn_model_file))          # the reference's own Keras HDF5 blob
    except Exception:  # noqa: BLE001
        pass
    for c in cand:
        if c.is_file():
            return c
    raise FileNotFoundError(
        "nn classifier weights not found: set GENOMAD_AMD_WEIGHTS to an .npz in the schema of "
        "genomad_amd/weights.py, or to the reference's nn_classifier.h5 (read through libhdf5, "
        "see genomad_amd/h5weights.py)")


def load_weights_file(path: Path) -> dict:
    """``.npz`` in the repo schema, or the reference's Keras HDF5 (genomad/data/nn_classifier.h5)."""
    from . import weights as W
    if Path(path).suffix.lower() in (".h5", ".hdf5"):
        from . import h5weights
        return h5weights.load_h5(path)
    return W.load_npz(path)


_ENGINE = None


def _engine():
    """One engine per process (the reference builds its Keras model twice, :309 and :379; here the
    weights are uploaded once and reused for the provirus pass)."""
    global _ENGINE
    if _ENGINE is None:
        from . import rccl
        from .engine import NNEngine
        # the reference module sets CUDA_VISIBLE_DEVICES=-1 at import (nn_classification.py:8), which
        # HIP honours; undo it before the HIP runtime is initialised (main() and install() do so as well)
        rccl.prepare_env()
        device = int(os.environ.get("GENOMAD_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _ENGINE = NNEngine(device, load_weights_file(find_weights()))
    return _ENGINE




def configured_precision() -> str:
    """GENOMAD_AMD_PRECISION if set (one of the library's arithmetic names), else the default; an unknown name is an error
    before anything is read or written, not a KeyError in the middle of a run."""
    from ._lib import OUT_OF_TOLERANCE, PRECISIONS
    name = os.environ.get("GENOMAD_AMD_PRECISION", DEFAULT_PRECISION)
    if name not in PRECISIONS:
        raise ValueError(f"GENOMAD_AMD_PRECISION={name!r}: expected one of {sorted(PRECISIONS)} (default {DEFAULT_PRECISION})")
    if name in OUT_OF_TOLERANCE and os.environ.get("GENOMAD_AMD_ALLOW_OUT_OF_TOLERANCE") != "1":
        raise ValueError(f"GENOMAD_AMD_PRECISION={name!r} is a measurement mode: max |dscore| {OUT_OF_TOLERANCE[name]:.1e} against the reference "
                         f"arithmetic on 10^6 windows, outside the 1e-4 tolerance; set GENOMAD_AMD_ALLOW_OUT_OF_TOLERANCE=1 to use it anyway")
    return name


def _warn_out_of_tolerance(console, name):
    """One log line per run for an arithmetic that was let through by GENOMAD_AMD_ALLOW_OUT_OF_TOLERANCE=1."""
    from ._lib import OUT_OF_TOLERANCE
    if name in OUT_OF_TOLERANCE and ("oot", name) not in _WARNED:
        _WARNED.add(("oot", name))
        msg = (f"WARNING: arithmetic {name} is OUTSIDE the 1e-4 score tolerance at scale (measured {OUT_OF_TOLERANCE[name]:.1e} on 10^6 "
               f"windows); it runs only because GENOMAD_AMD_ALLOW_OUT_OF_TOLERANCE=1 is set.  The 64-window parity sentinel cannot see a 1-in-10^5 tail.")
        (console.log if console is not None else print)(msg)
_WARNED = set()


# what to recompute a batch with when an f16-operand arithmetic returns non-finite scores: the Toom-Cook form's transformed
# activations leave the f16 range first (|activation| > ~2 000), the direct f16 forms at 65 504, bf16x3 has the f32 range
RANGE_FALLBACKS = {"f16x3tk": ("f16x3", "bf16x3"), "f16x3tc": ("f16x3", "bf16x3"), "f16x3": ("bf16x3",), "f16c6": ("bf16x3",)}

# ---- the k-mer tables of "f16x3tk" (gnn_fused_tk.hip): conv2 as a 137 GB table of all 14-mers, head A's pair products as an 8.8 GB
# (entry, 9-mer) table.  1.4x the default arithmetic's speed, the same scores to 1e-6 - and 1.5 .. 6 seconds of hipMalloc to set up,
# which a window costing 5 us repays after a few million windows.  GENOMAD_AMD_KMER_TABLES: "auto" (default) builds them when the
# run's input is at least GENOMAD_AMD_KMER_TABLES_MIN_GB (default 18 = 3 M windows) of FASTA or the engine already has them,
# "1" always, "0" never.  A device that cannot hold them (156 GB + workspaces) keeps the default arithmetic; so does the whole
# run when ANY rank cannot (the scores must not depend on which rank classified a contig).  The decision depends on the input,
# never on the number of ranks: a run gives the same bits on 1 GPU and on 8.
KMER_TABLES_MIN_GB = 18.0


def select_arithmetic(eng, precision: str, input_bytes: int, comm=None, console=None) -> str:
    """The arithmetic main() classifies with: ``precision`` (configured_precision()), promoted from "f16x3tc" to "f16x3tk" when the
    k-mer tables are there or worth building (see above).  An explicit GENOMAD_AMD_PRECISION=f16x3tk that cannot be served is an
    error; every other arithmetic is returned as it is."""
    if precision not in ("f16x3tc", "f16x3tk") or not hasattr(eng, "build_kmer_tables"):
        return precision
    policy = os.environ.get("GENOMAD_AMD_KMER_TABLES", "auto").lower()
    if policy not in ("auto", "0", "1"):
        raise ValueError(f"GENOMAD_AMD_KMER_TABLES={policy!r}: expected auto, 0 or 1")
    explicit = precision == "f16x3tk"
    min_bytes = float(os.environ.get("GENOMAD_AMD_KMER_TABLES_MIN_GB", KMER_TABLES_MIN_GB)) * 1e9
    want = explicit or policy == "1" or (policy == "auto" and (eng.has_kmer_tables() or input_bytes >= min_bytes))
    if comm is not None:                       # eng.has_kmer_tables() is per process: every rank follows what ANY rank wants
        want = bool(np.asarray(comm.allgather_i64([1 if want else 0])).any())
    ok = False
    if want:
        t = time.perf_counter()
        had = eng.has_kmer_tables()
        ok = eng.build_kmer_tables()
        if ok and not had and console is not None:
            console.log(f"k-mer tables built on the device in {time.perf_counter() - t:.1f} s (14-mer table of conv2, 9-mer table of head A's pair products).")
    if comm is not None:                       # ... and what EVERY rank could build (allgather_i64: the transport every comm of sharding.py has)
        ok = bool(np.asarray(comm.allgather_i64([1 if ok else 0])).all())
    if want and not ok:
        if explicit:
            raise RuntimeError("GENOMAD_AMD_PRECISION=f16x3tk: the device cannot hold the k-mer tables (156 GB + workspaces); "
                               "unset it to run the default arithmetic")
        if console is not None:
            console.log("k-mer tables not built (device memory): classifying with the default arithmetic f16x3tc.")
    return "f16x3tk" if ok else "f16x3tc"


def _range_fallback(console, what, to):
    """The f16-operand modes cannot represent activations beyond their range; scores that come back
    non-finite are recomputed with the next arithmetic of RANGE_FALLBACKS.  ``what`` is the arithmetic that just
    returned the non-finite scores (on the second hop of a chain: the first fallback, not the configured one).
    Said once per run and pair."""
    if (what, to) not in _WARNED:
        _WARNED.add((what, to))
        msg = (f"Non-finite class scores from the f16 arithmetic ({what}): activations left its range; "
               f"recomputing the affected batch with {to}.")
        (console.log if console is not None else print)(msg)


def _with_range_fallback(run, precision, console=None):
    """``run(arithmetic)`` -> (scores, extra); walks RANGE_FALLBACKS[precision] while the scores are non-finite.
    Returns (scores, extra, arithmetic that produced them)."""
    used = precision
    pr, extra = run(used)
    for nxt in RANGE_FALLBACKS.get(precision, ()):
        if np.isfinite(pr).all():
            break
        _range_fallback(console, used, nxt)
        used = nxt
        pr, extra = run(used)
    return pr, extra, used


def classify_contigs_safely(eng, seq, offsets, single_window, precision, console=None):
    """NNEngine.classify_contigs with the range fallback of :func:`_range_fallback`."""
    pr, wid, _ = _with_range_fallback(lambda a: eng.classify_contigs(seq, offsets, single_window, a), precision, console)
    return pr, wid


# ---- runtime parity sentinel ---------------------------------------------------------------------
# The 1e-4 claim of the production arithmetic has only ever met synthetic weights (the trained nn_classifier.h5 is not in the
# checkout).  So every run of main() classifies its first windows a second time with the exact-f32 device path (GNN_PREC_F32:
# unfused f32 FMA kernels, the parity anchor of tests/ and bench.py) and compares: the only parity evidence a user with the real
# weights ever gets, and what turns the range / precision assumptions of the f16 limb arithmetic (activations neither beyond
# ~2 000 nor so small that the low limbs go subnormal) into checked ones.  Cost: one 64-window launch of the f32 path, 6 ms
# including its 0.6 GB activation workspace (tests/test_gpu_parity.py::test_parity_sentinel_... prints it).  GENOMAD_AMD_NO_SENTINEL=1 opts out.
SENTINEL_WINDOWS = 64
SENTINEL_TOL = 1e-4            # BASELINE.json north_star: per-class scores within 1e-4 absolute of the reference path


def sentinel_windows(seq, offsets, single_window, limit=SENTINEL_WINDOWS) -> np.ndarray:
    """The first ``limit`` candidate windows of a packed contig buffer as (k, 6000) upper-cased, N-padded bytes
    (nn_classification.py:68-72; the N-content rule is irrelevant for a parity sample)."""
    starts, lens, _, _ = sequence.candidate_spans(np.asarray(offsets, np.int64), single_window)
    k = int(min(len(starts), limit))
    win = np.full((k, sequence.WINDOW), ord("N"), dtype=np.uint8)
    for i in range(k):
        w = np.asarray(seq[int(starts[i]):int(starts[i]) + int(lens[i])], dtype=np.uint8)
        win[i, :len(w)] = np.where((w >= 97) & (w <= 122), w - 32, w)          # sequence.py:35-36 upper()
    return win


def parity_sentinel(eng, windows, precision, console=None):
    """max |dscore| of the production arithmetic (after its range fallbacks) against GNN_PREC_F32 on ``windows``;
    None when switched off, for the exact arithmetic itself, or without windows.  Logs one line."""
    if os.environ.get("GENOMAD_AMD_NO_SENTINEL") == "1" or precision == "f32" or not len(windows):
        return None
    from ._lib import GnnError
    log = console.log if console is not None else print
    exact = None
    for k in (len(windows), 16, 4):          # the exact path keeps 3 x 3 MB of f32 activations per window: on a short device try fewer
        try:
            exact = eng.classify(windows[:k], "f32")
            windows = windows[:k]
            break
        except GnnError as exc:              # the production arithmetic may well fit where the exact one does not (shared / partitioned GPUs)
            if "hipMalloc" not in str(exc) and "memory" not in str(exc).lower():
                raise
    if exact is None:
        log("Parity sentinel skipped: no device memory for the exact-f32 path (4 windows need 37 MB of activations).")
        return None
    got, _, used = _with_range_fallback(lambda a: (eng.classify(windows, a), None), precision, console)
    d = float(np.abs(got.astype(np.float64) - exact).max()) if np.isfinite(got).all() and np.isfinite(exact).all() else float("inf")
    log(f"Parity sentinel: max |dscore| of {used} against the exact-f32 path on the first {len(windows)} windows = {d:.2e} "
        f"(tolerance {SENTINEL_TOL:.0e}).")
    return d


def sentinel_verdict(comm, d, console, precision):
    """Every rank brings its own sentinel result (None = nothing to check); all ranks leave together if any failed."""
    bad = d is not None and not d <= SENTINEL_TOL
    if comm is not None:
        bad = bool(np.asarray(comm.allgather_i64([int(bad)]))[:, 0].any())
    if bad:
        console.error(
            f"Parity sentinel FAILED: the {precision} arithmetic differs from the exact-f32 device path by more than {SENTINEL_TOL:.0e} "
            "on this run's first windows (weights whose activations leave the range the f16 limb arithmetic was validated for). "
            "Re-run with GENOMAD_AMD_PRECISION=bf16x3 (f32 range) or GENOMAD_AMD_PRECISION=f32 (exact, slow); "
            "GENOMAD_AMD_NO_SENTINEL=1 disables this check.")
        sys.exit(1)


class GpuBackend:
    """Scores windows and averages them per contig on the GPU (libgenomad_nn_hip.so)."""

    def __init__(self, batch_size: int, console=None):
        self.eng = _engine()
        self.chunk = max(int(batch_size), 4096)
        self.precision = configured_precision()
        if (self.precision == "f16x3tc" and getattr(self.eng, "has_kmer_tables", lambda: False)()
                and os.environ.get("GENOMAD_AMD_KMER_TABLES", "auto") != "0"):
            self.precision = "f16x3tk"          # an engine that already holds the tables (a long-lived process) uses them
        self.console = console
        self.sentinel = None          # max |dscore| of the first windows this backend scored (parity_sentinel)
        self._sentinel_done = False

    def score(self, windows: np.ndarray) -> np.ndarray:
        if not self._sentinel_done:
            self._sentinel_done = True
            self.sentinel = parity_sentinel(self.eng, windows[:SENTINEL_WINDOWS], self.precision, self.console)
        out = []
        for a in range(0, len(windows), self.chunk):
            s, _, _ = _with_range_fallback(lambda p, a=a: (self.eng.classify(windows[a:a + self.chunk], p), None),
                                           self.precision, self.console)
            out.append(s)
        return np.concatenate(out) if out else np.zeros((0, 3), np.float32)

    def segment_mean(self, scores, ids, n_segments) -> np.ndarray:
        return self.eng.segment_mean(scores, ids, n_segments)


def classify_windows(windows: np.ndarray, contig_ids: np.ndarray, n_contigs: int, backend, comm=None):
    """Window scores -> per-contig mean (nn_classification.py:316-320), sharded over the ranks of ``comm``
    (rank 0 gets the result, other ranks None)."""
    from . import sharding
    scores = sharding.classify_sharded(windows, backend.score, comm)
    if scores is None:
        return None
    return backend.segment_mean(scores, contig_ids, n_contigs)


def _encode(fasta_path, enc_dir: Path, window_id_path: Path, single_window: bool, names_key: str, ids_key: str):
    names, ids, windows = sequence.encode_fasta(fasta_path, single_window)
    np.save(enc_dir / f"{len(windows)}.win.npy", windows)
    np.savez_compressed(window_id_path, **{names_key: names, ids_key: ids})
    return names, ids, windows


def main(input_path, output_path, single_window, batch_size, restart, threads, verbose, cleanup,
         _backend=None, _comm=None):
    """``_backend`` (tests only) replaces the GPU engine with an object offering score() and
    segment_mean(); ``_comm`` (tests only) replaces the RCCL transport with another implementation of the
    sharding.py transport interface.  The product path always builds a :class:`GpuBackend` and an
    :class:`genomad_amd.rccl.RcclComm`, and fails without a GPU."""
    from . import rccl, sharding
    rccl.prepare_env()        # before ANY HIP call: drops the reference's CUDA_VISIBLE_DEVICES=-1 (:8)
    configured_precision()    # a mistyped GENOMAD_AMD_PRECISION stops here
    input_path, output_path = Path(input_path), Path(output_path)
    if _comm is not None:
        comm = _comm
    elif _backend is None and rccl.world_from_env()[1] > 1:
        comm = rccl.comm_for(_engine())      # one process per GPU: RCCL communicator of this rank's engine
    else:
        comm = None
    rank, world = (comm.rank, comm.world) if comm is not None else (0, 1)
    rank0 = rank == 0
    if not output_path.is_dir():
        output_path.mkdir(exist_ok=True)
    prefix = sequence.prefix_of(input_path)
    outputs = Outputs(prefix, output_path)
    console = Console(output_file=outputs.nn_classification_log if rank0 else None, verbose=verbose and rank0)
    parameter_dict = {"single_window": single_window}
    md5_async(input_path)            # starts hashing now; check_fasta and the stages overlap with it
    device_front_end = _backend is None and os.environ.get("GENOMAD_AMD_FRONT_END", "device") == "device"

    def everywhere(*flags):
        """Decisions read from files that rank 0 is about to rewrite are taken on rank 0 and made known to
        the other ranks, so that all of them walk through the same gathers."""
        return [bool(f) for f in sharding.broadcast_flags(comm, flags)] if comm is not None else [bool(f) for f in flags]

    (classify_proviruses,) = everywhere(check_provirus_execution(outputs, input_path) if rank0 else False)
    output_files = [outputs.nn_classification_execution_info, outputs.encoded_sequences_dir,
                    outputs.nn_classification_output, outputs.nn_classification_npz_output]
    if classify_proviruses:
        output_files += [outputs.encoded_proviruses_dir, outputs.provirus_nn_classification_output,
                         outputs.provirus_nn_classification_npz_output]
    console.log(f"Executing geNomad nn-classification (genomad_amd, MI355X). Outputs in {outputs.nn_classification_dir}.")

    def fail_on_bad_fasta(ok: bool):                                           # :164-170
        if not ok:
            console.error(f"{input_path} is either empty or contains multiple entries with the same identifier. "
                          "Please check your input FASTA file and execute genomad nn-classification again.")
            sys.exit(1)

    # The FASTA is validated once, on rank 0 (in bounded memory: sequence.check_fasta), and the verdict is
    # shared.  The device front end runs it on a helper thread while the GPU already classifies; nothing is
    # written (no outputs, no execution info) before ``gate()`` has seen the verdict, so an invalid input
    # leaves the same state behind as in the reference, which checks first.
    skip, changed = False, False                                               # :175-197
    if rank0 and (outputs.nn_classification_execution_info.exists() and any(p.exists() for p in output_files)
                  and not restart):
        skip = compare_executions(input_path, parameter_dict, outputs.nn_classification_execution_info)
        changed = not skip
    skip, changed = everywhere(skip, changed)
    # With several ranks the validation is sharded like the classification: every rank collects the accessions of the records
    # of ITS share while it reads it (no second pass over the file), and one gather of 64-bit digests decides
    # (sharding.fasta_verdict).  Only when the main stage really reads the file: a resumed run that finds the scores on disk
    # validates as before, on rank 0.
    (sharded_check,) = everywhere(device_front_end and world > 1 and rank0
                                  and not (skip and outputs.nn_classification_npz_output.exists()))
    check_pool = concurrent.futures.ThreadPoolExecutor(max_workers=1) if (device_front_end and rank0 and not sharded_check) else None
    check_future = check_pool.submit(sequence.check_fasta, input_path) if check_pool else None
    if not device_front_end:
        fail_on_bad_fasta(everywhere(sequence.check_fasta(input_path) if rank0 else True)[0])
    if skip:
        console.log("Previous execution detected. Steps will be skipped unless their outputs are not found. "
                    "Use the --restart option to force the execution of all the steps again.")
    elif changed:
        console.log("The input file or the parameters changed since the last execution. "
                    "Previous outputs will be overwritten.")
    state = {"info_writer": None, "gated": False, "verdict": None}
    start_time = datetime.now(timezone.utc).astimezone().isoformat()

    def gate():
        """First call (every rank, at the same point of the program): wait for the FASTA verdict, then start
        writing the execution info (as soon as the digest is ready; a non-daemon thread, so it also
        completes if a stage ends the run with sys.exit).  Every writer of an output file calls this first."""
        if state["gated"]:
            return
        state["gated"] = True
        if device_front_end and sharded_check:
            if state["verdict"] is None:                         # gate() before the main stage ran fasta_verdict: a bug, not a bad FASTA
                raise RuntimeError("internal error: gate() was reached before the sharded FASTA verdict was computed")
            fail_on_bad_fasta(state["verdict"])                  # decided by all ranks together in the main stage
        elif device_front_end:
            fail_on_bad_fasta(everywhere(check_future.result() if rank0 else True)[0])
        if rank0:
            outputs.nn_classification_dir.mkdir(exist_ok=True)
            state["info_writer"] = threading.Thread(
                target=write_execution_info, name="genomad-amd-execution-info",
                args=(MODULE_NAME, input_path, parameter_dict, outputs.nn_classification_execution_info, start_time))
            state["info_writer"].start()

    if not device_front_end:
        gate()

    def stage(fasta, enc_dir, wid_path, npz_path, tsv_path, names_key, ids_key, what):
        windows = None
        (have_enc,) = everywhere(rank0 and skip and wid_path.exists() and bool(
            len(list(enc_dir.glob("*.win.npy"))) or len(list(enc_dir.glob("*.tfrec")))))       # :215-225
        if have_enc:
            console.log(f"{enc_dir.name} was found. Skipping {what} encoding.")
            z = np.load(wid_path)
            names, ids = z[names_key], z[ids_key]
        else:
            if rank0:
                if enc_dir.is_dir():
                    shutil.rmtree(enc_dir)
                enc_dir.mkdir()
                names, ids, windows = _encode(fasta, enc_dir, wid_path, single_window, names_key, ids_key)
            else:
                names, ids, windows = sequence.encode_fasta(fasta, single_window)
            console.log(f"Encoded {what} data written to {enc_dir.name}.")
        (have_npz,) = everywhere(rank0 and skip and npz_path.exists())                   # :284-292
        predictions = None
        if have_npz:
            console.log(f"{npz_path.name} was found. Skipping {what} classification.")
            if rank0:
                z = np.load(npz_path)
                names, predictions = z[names_key], z["predictions"]
        else:
            if windows is None:
                files = sorted(enc_dir.glob("*.win.npy"))
                if files:
                    windows = np.concatenate([np.load(f) for f in files])
                else:
                    # a directory encoded by the reference itself (TFRecords of tokens, :43-52): rebuild
                    # windows that tokenise to exactly those tokens
                    from . import tfrecord
                    windows = tfrecord.tokens_to_bases(tfrecord.read_dir(enc_dir))
            if not len(windows):                                                     # :297-299
                console.error("No sequences were found. Please check your input FASTA.")
                sys.exit(1)
            backend = _backend if _backend is not None else GpuBackend(batch_size, console)
            predictions = classify_windows(windows, ids, len(names), backend, comm)
            if _backend is None:
                sentinel_verdict(comm, getattr(backend, "sentinel", None), console, getattr(backend, "precision", "?"))
            console.log(f"{what.capitalize()}s classified.")
            if rank0:
                np.savez_compressed(npz_path, **{names_key: names, "predictions": predictions})   # :326-330
        if cleanup and rank0 and enc_dir.is_dir():                                   # :335-337
            console.log(f"Deleting encoded {what} data.")
            shutil.rmtree(enc_dir)
        if rank0:
            write_tsv(tsv_path, names, predictions)                                  # :340-352 (always rewritten)

    def stage_device(fasta, enc_dir, wid_path, npz_path, tsv_path, names_key, ids_key, what):
        """Product path: the contig front end (NNEngine.classify_contigs) does windowing, the N rule,
        tokenising, classification and the per-contig mean on the GPU, so encoding and classification
        are one step; ``<prefix>_seq_window_id.npz`` is still written.  With several ranks the CONTIGS are
        sharded: every rank reads and packs only its own record-aligned byte range of the file (for
        compressed inputs: every world-th record-aligned chunk of the decompressed stream), classifies it
        on its GPU, and rank 0 collects the per-contig scores with one gather and writes the files —
        results are bit-identical for any number of ranks."""
        (have_npz,) = everywhere(rank0 and skip and npz_path.exists())                   # :284-292
        names = predictions = None
        if have_npz:
            gate()
            console.log(f"{npz_path.name} was found. Skipping {what} classification.")
            if rank0:
                z = np.load(npz_path)
                names, predictions = z[names_key], z["predictions"]
        else:
            precision = configured_precision()
            _warn_out_of_tolerance(console, precision)
            eng = _engine()
            size = Path(fasta).stat().st_size * (1 if sequence.compression_of(fasta) == "uncompressed" else 4)
            precision = select_arithmetic(eng, precision, size, comm, console)
            parts = []
            sentinel = {"d": None, "done": False}

            def classify(sq, off):
                if not sentinel["done"] and len(off) > 1:           # this rank's first piece with a contig: the run's parity sample
                    sentinel["done"] = True
                    sentinel["d"] = parity_sentinel(eng, sentinel_windows(sq, off, single_window), precision, console)
                    if comm is None:                                # one rank: a failed check stops the run here, not after the whole
                        sentinel_verdict(None, sentinel["d"], console, precision)       # file (several ranks leave together below)
                return classify_contigs_safely(eng, sq, off, single_window, precision, console)

            validate = sharded_check and fasta is input_path        # the provirus FASTA is geNomad's own output: never validated
            seen = []                                               # digests of the accessions of ALL records of this rank's share (validate)

            def pack(text):
                if validate:                                        # one native pass, before the in-place pack consumes the text
                    seen.append(sequence.accession_digests_of_text(text))
                return sequence.pack_text(text, strip_n=True)

            if sequence.compression_of(fasta) == "uncompressed":
                # this rank's record-aligned share of the file, in pieces of about 128 MB: piece k+1
                # is read and packed on a helper thread while the GPU classifies piece k
                share = Path(fasta).stat().st_size // world
                pieces = int(min(64, max(1, -(-share // (128 << 20)))))
                read = lambda k: pack(sequence._read_text_array(  # noqa: E731
                    fasta, sequence.record_aligned_range(fasta, rank, world, k, pieces)))
                with concurrent.futures.ThreadPoolExecutor(max_workers=1) as pool:
                    nxt = pool.submit(read, 0)
                    for k in range(pieces):
                        nm, sq, off = nxt.result()
                        if k + 1 < pieces:
                            nxt = pool.submit(read, k + 1)
                        pr, wid = classify(sq, off)
                        parts.append((rank * 64 + k, nm, pr, wid))
            else:
                # compressed streams cannot be read by byte range: every rank decompresses the stream
                # chunk by chunk (bounded memory) and packs + classifies every world-th chunk
                for i, chunk in enumerate(sequence.iter_text_chunks(fasta)):
                    if i % world != rank:
                        continue
                    nm, sq, off = pack(chunk)
                    pr, wid = classify(sq, off)
                    parts.append((i, nm, pr, wid))
            if validate:
                digests = np.concatenate(seen) if seen else np.zeros(0, dtype="<u8")
                state["verdict"] = sharding.fasta_verdict(comm, digests, lambda: sequence.check_fasta(input_path))
            sentinel_verdict(comm, sentinel["d"], console, precision)      # before anything is written
            names, predictions, ids, n_windows = sharding.gather_contig_parts(comm, parts)
            gate()
            if not n_windows:                                                        # :297-299
                console.error("No sequences were found. Please check your input FASTA.")
                sys.exit(1)
            if rank0:
                if enc_dir.is_dir():
                    shutil.rmtree(enc_dir)
                enc_dir.mkdir()
                np.savez_compressed(wid_path, **{names_key: names, ids_key: ids})
                console.log(f"{what.capitalize()}s classified ({len(ids)} windows).")
                np.savez_compressed(npz_path, **{names_key: names, "predictions": predictions})
        if cleanup and rank0 and enc_dir.is_dir():
            console.log(f"Deleting encoded {what} data.")
            shutil.rmtree(enc_dir)
        if rank0:
            write_tsv(tsv_path, names, predictions)

    run = stage_device if device_front_end else stage
    try:
        run(input_path, outputs.encoded_sequences_dir, outputs.seq_window_id_output,
            outputs.nn_classification_npz_output, outputs.nn_classification_output,
            "contig_names", "contig_ids", "sequence")
        if classify_proviruses:                                                      # :248-281, :355-425
            run(outputs.find_proviruses_nucleotide_output, outputs.encoded_proviruses_dir,
                outputs.provirus_window_id_output, outputs.provirus_nn_classification_npz_output,
                outputs.provirus_nn_classification_output, "provirus_names", "provirus_ids", "provirus")
    finally:
        if check_pool is not None:
            check_pool.shutdown(wait=True)
        if state["info_writer"] is not None:
            if os.environ.get("GENOMAD_AMD_DEFER_EXECUTION_INFO") == "1" and sys.exc_info()[0] is None:
                # opt-in: do not wait for the input's md5 (one sequential pass at ~1 GB/s, utils.py:216-223 - with 8 GPUs it is
                # the longest thing main() does).  The scores are on disk; the execution-info JSON appears, with the same
                # content, when the digest is ready (a non-daemon thread: the process does not exit before).  A caller that
                # runs the next geNomad module in the same process right away (end-to-end: aggregated-classification requires
                # the JSON, aggregated_classification.py:101) must call wait_execution_info() first - hence opt-in.
                _DEFERRED.append(state["info_writer"])
            else:
                state["info_writer"].join()
    if comm is not None:
        comm.barrier()               # rank 0 has written everything before any rank returns
    console.log("geNomad nn-classification finished!")


_DEFERRED = []


def wait_execution_info():
    """Block until every execution-info JSON deferred by GENOMAD_AMD_DEFER_EXECUTION_INFO=1 has been written."""
    while _DEFERRED:
        _DEFERRED.pop().join()


def install():
    """Make the reference CLI use this module: ``genomad.nn_classification`` is looked up at call
    time (cli.py:772, :1367), so rebinding the attribute is enough."""
    from . import rccl
    import genomad
    import genomad.modules          # importing the reference module exports CUDA_VISIBLE_DEVICES=-1 (:8) ...
    rccl.prepare_env()              # ... which HIP would honour: undo it before any HIP call
    this = sys.modules[__name__]
    genomad.nn_classification = this
    genomad.modules.nn_classification = this
    sys.modules["genomad.modules.nn_classification"] = this
    return this
