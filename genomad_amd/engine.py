"""NNEngine — thin host wrapper over the C ABI: one engine = one gnn_ctx = one GPU.

Mirrors what the reference does with a Keras model inside
genomad/modules/nn_classification.py:309-320: build, load weights, predict batches,
segment-mean per contig.  All arithmetic happens in libgenomad_nn_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib, weights as _weights
from ._lib import GnnError, check  # noqa: F401  (re-export)


class DeviceBuffer:
    """A raw device allocation owned by an engine (no torch needed for buffers)."""

    def __init__(self, engine, nbytes: int):
        self.engine, self.nbytes = engine, int(nbytes)
        p = C.c_void_p()
        check(engine.lib.gnn_dev_alloc(engine.ctx, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(self.engine.lib.gnn_memcpy_h2d(self.engine.ctx, self.ptr, arr.ctypes.data, arr.nbytes))

    def download(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(self.engine.lib.gnn_memcpy_d2h(self.engine.ctx, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            check(self.engine.lib.gnn_dev_free(self.engine.ctx, self.ptr))
            self.ptr = None


class NNEngine:
    def __init__(self, device: int = 0, weights: dict = None, chunk: int = None):
        self.lib = _lib.load()
        ctx = C.c_void_p()
        check(self.lib.gnn_create(int(device), C.byref(ctx)))
        self.ctx = ctx
        self.device = int(device)
        self._weights_keepalive = None
        if chunk:
            check(self.lib.gnn_set_chunk(self.ctx, int(chunk)))
        if weights is not None:
            self.load_weights(weights)

    # -- life cycle ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "ctx", None):
            self.lib.gnn_destroy(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pci_bus_id(self) -> str:
        buf = C.create_string_buffer(32)
        check(self.lib.gnn_device_pci_bus_id(self.ctx, buf, 32))
        return buf.value.decode()

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cus, mem = C.c_int(), C.c_int64()
        check(self.lib.gnn_device_info(self.ctx, name, 256, C.byref(cus), C.byref(mem)))
        return {"name": name.value.decode(), "cus": cus.value, "hbm_bytes": mem.value}

    def mem_info(self):
        """(free, total) bytes of the engine's device."""
        f, t = C.c_int64(), C.c_int64()
        check(self.lib.gnn_device_mem_info(self.ctx, C.byref(f), C.byref(t)))
        return f.value, t.value

    def load_weights(self, weights: dict):
        w = _weights.validate(weights)
        s, keep = _weights.to_struct(w)
        check(self.lib.gnn_load_weights(self.ctx, C.byref(s)))
        self._weights_keepalive = keep

    def build_kmer_tables(self, reserve_bytes: int = -1) -> bool:
        """The k-mer tables of "f16x3tk" (156 GB: x2 per 14-mer, head A's pair products per (entry, 9-mer), conv2's tap tables and x1 over head A's index space); False - with nothing
        allocated - on a device that cannot hold them behind `reserve_bytes` (< 0: the library's default reserve)."""
        rc = self.lib.gnn_build_kmer_tables(self.ctx, int(reserve_bytes))
        if rc == _lib.ERR_NOMEM:
            return False
        check(rc)
        return True

    def has_kmer_tables(self) -> bool:
        return bool(self.lib.gnn_has_kmer_tables(self.ctx))

    def drop_kmer_tables(self):
        check(self.lib.gnn_drop_kmer_tables(self.ctx))

    def sync(self):
        check(self.lib.gnn_sync(self.ctx))

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    # -- hot path -----------------------------------------------------------------------
    @staticmethod
    def _check_bases(bases) -> np.ndarray:
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        if b.ndim != 2 or b.shape[1] != _lib.WINDOW:
            raise ValueError(f"bases must have shape (n, {_lib.WINDOW}), got {b.shape}")
        return b

    def tokenize(self, bases) -> np.ndarray:
        """(n,6000) uint8 padded upper-case windows -> (n,5997) uint16 tokens (sequence.py:170-193)."""
        b = self._check_bases(bases)
        out = np.empty((len(b), _lib.TOKENS), dtype=np.uint16)
        check(self.lib.gnn_tokenize(self.ctx, b.ctypes.data, len(b), out.ctypes.data))
        return out

    def onehot(self, bases, dtype="u8") -> np.ndarray:
        """Stand-alone encoder: (n,6000) bases -> (n,5997,257) one-hot (model.py:9-11)."""
        b = self._check_bases(bases)
        code, npdt = {"u8": (_lib.OH_U8, np.uint8), "bf16": (_lib.OH_BF16, np.uint16),
                      "f32": (_lib.OH_F32, np.float32)}[dtype]
        shape = (len(b), _lib.TOKENS, _lib.DEPTH)
        nbytes = int(np.prod(shape)) * np.dtype(npdt).itemsize
        db, do = self.alloc(max(b.nbytes, 1)), self.alloc(max(nbytes, 1))
        try:
            if len(b):
                db.upload(b)
            check(self.lib.gnn_onehot_dev(self.ctx, db.ptr, len(b), code, do.ptr))
            self.sync()
            return do.download(shape, npdt)
        finally:
            db.free()
            do.free()

    def classify(self, bases, precision=_lib.DEFAULT_PRECISION) -> np.ndarray:
        """(n,6000) uint8 windows -> (n,3) float32 class scores (chromosome, plasmid, virus)."""
        b = self._check_bases(bases)
        out = np.empty((len(b), _lib.CLASSES), dtype=np.float32)
        check(self.lib.gnn_classify(self.ctx, b.ctypes.data, len(b), _lib.PRECISIONS[precision],
                                    out.ctypes.data))
        return out

    def classify_dev(self, bases_ptr: int, n: int, scores_ptr: int, precision=_lib.DEFAULT_PRECISION):
        """Asynchronous: device pointers in and out, enqueued on the engine's stream."""
        check(self.lib.gnn_classify_dev(self.ctx, bases_ptr, int(n), _lib.PRECISIONS[precision],
                                        scores_ptr))

    def classify_dev_async(self, bases_ptr: int, n: int, scores_ptr: int, precision=_lib.DEFAULT_PRECISION):
        """classify_dev whose last back end may still run beside the next call's front end; call :meth:`flush`
        (or sync / download / a collective) before reading the scores."""
        check(self.lib.gnn_classify_dev_async(self.ctx, bases_ptr, int(n), _lib.PRECISIONS[precision], scores_ptr))

    def flush(self):
        check(self.lib.gnn_classify_flush(self.ctx))

    def debug_forward(self, bases, precision="f32", taps=("m_a", "m_b", "yp_a", "yp_b",
                                                         "alpha_a", "alpha_b", "feat")):
        """Scores plus the requested intermediates as a dict of numpy arrays."""
        b = self._check_bases(bases)
        n = len(b)
        shapes = {"x1": (n, _lib.TOKENS, _lib.CH), "x2": (n, _lib.TOKENS, _lib.CH),
                  "x3": (n, _lib.TOKENS, _lib.CH), "m_a": (n, _lib.PATCHES), "m_b": (n, _lib.PATCHES),
                  "yp_a": (n, _lib.POOLED, _lib.CH), "yp_b": (n, _lib.POOLED, _lib.CH),
                  "alpha_a": (n, _lib.POOLED), "alpha_b": (n, _lib.POOLED), "feat": (n, _lib.FEAT)}
        arrays = {k: np.empty(shapes[k], dtype=np.float32) for k in taps}
        t = _lib.Taps(**{k: v.ctypes.data_as(C.POINTER(C.c_float)) for k, v in arrays.items()})
        scores = np.empty((n, _lib.CLASSES), dtype=np.float32)
        check(self.lib.gnn_debug_forward(self.ctx, b.ctypes.data, n, _lib.PRECISIONS[precision],
                                         scores.ctypes.data, C.byref(t)))
        return scores, arrays

    def classify_contigs(self, seq: np.ndarray, offsets: np.ndarray, single_window: bool = False,
                         precision=_lib.DEFAULT_PRECISION):
        """Contig front end: packed raw contig bytes -> per-contig scores.

        Does what generate_data + the predict loop + segment_mean do (nn_classification.py:54-82,
        :316-320) in ONE library call (``gnn_classify_contigs``): candidate windows are cut in native code,
        the packed buffer goes up in pieces on a copy stream while earlier pieces are classified, the
        N-content rule, upper-casing, padding, tokenising, classification and the per-contig mean run on the
        device and the window scores never leave it.  Returns (contig_scores (n_contigs, 3), contig ids of
        the kept windows).
        """
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        return self._classify_contigs(seq.ctypes.data, 1, seq.nbytes, offsets, single_window, precision)

    def classify_contigs_dev(self, seq_ptr: int, offsets: np.ndarray, single_window: bool = False,
                             precision=_lib.DEFAULT_PRECISION):
        """Same as :meth:`classify_contigs` for a packed contig buffer that is already resident in
        HBM (``seq_ptr`` = device address of byte 0, ``offsets`` = (n_contigs+1,) byte offsets)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        return self._classify_contigs(seq_ptr, 0, int(offsets[-1]) if len(offsets) else 0, offsets, single_window, precision)

    def _classify_contigs(self, seq_ptr, on_host, seq_bytes, offsets, single_window, precision):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if offsets.ndim != 1 or len(offsets) < 1:
            raise ValueError("offsets must hold n_contigs + 1 byte offsets")
        n_contigs = len(offsets) - 1
        scores = np.zeros((n_contigs, _lib.CLASSES), dtype=np.float32)
        cap = int(((np.diff(offsets) + _lib.WINDOW - 1) // _lib.WINDOW).sum()) if n_contigs else 0
        ids = np.empty(max(cap, 1), dtype=np.int64)
        n = C.c_int64()
        check(self.lib.gnn_classify_contigs(self.ctx, seq_ptr, int(on_host), int(seq_bytes), offsets.ctypes.data, n_contigs,
                                            int(bool(single_window)), _lib.PRECISIONS[precision], scores.ctypes.data,
                                            ids.ctypes.data, cap, C.byref(n)))
        return scores, ids[:n.value].copy()

    def classify_contigs_spans(self, seq_ptr: int, offsets: np.ndarray, single_window: bool = False,
                               precision=_lib.DEFAULT_PRECISION):
        """The same result assembled on the host from the span-level entry points (gnn_span_byte_count,
        gnn_classify_spans, gnn_segment_mean) — kept as an independently coded cross-check for the tests."""
        from . import sequence as S
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n_contigs = len(offsets) - 1
        starts, lens, ids, window_n = S.candidate_spans(offsets, single_window)
        if not len(starts):
            return np.zeros((n_contigs, _lib.CLASSES), np.float32), ids
        later = np.flatnonzero(window_n > 0)              # window 0 is never skipped (:70)
        keep = np.ones(len(starts), dtype=bool)
        if len(later):
            st, ln = np.ascontiguousarray(starts[later]), np.ascontiguousarray(lens[later])
            counts = np.empty(len(later), dtype=np.int32)
            check(self.lib.gnn_span_byte_count(self.ctx, seq_ptr, st.ctypes.data, ln.ctypes.data, len(later),
                                               ord("N"), counts.ctypes.data))
            keep[later[counts > S.MAX_N]] = False
        starts, lens, ids = (np.ascontiguousarray(a[keep]) for a in (starts, lens, ids))
        scores = np.empty((len(starts), _lib.CLASSES), dtype=np.float32)
        check(self.lib.gnn_classify_spans(self.ctx, seq_ptr, starts.ctypes.data, lens.ctypes.data, len(starts),
                                          _lib.PRECISIONS[precision], scores.ctypes.data))
        return self.segment_mean(scores, ids, n_contigs), ids

    def segment_mean(self, scores, ids, n_segments=None) -> np.ndarray:
        """tf.math.segment_mean(scores, ids) (nn_classification.py:320); ids sorted ascending."""
        s = np.ascontiguousarray(scores, dtype=np.float32)
        i = np.ascontiguousarray(ids, dtype=np.int64)
        if n_segments is None:
            n_segments = int(i.max()) + 1 if len(i) else 0
        out = np.zeros((n_segments, _lib.CLASSES), dtype=np.float32)
        check(self.lib.gnn_segment_mean(self.ctx, s.ctypes.data, i.ctypes.data, len(i), n_segments,
                                        out.ctypes.data))
        return out

    def synth_windows_dev(self, first: int, n: int, bases_ptr: int, seed: int = 1234):
        check(self.lib.gnn_synth_windows_dev(self.ctx, seed, int(first), int(n), bases_ptr))

    def synth_windows(self, first: int, n: int, seed: int = 1234) -> np.ndarray:
        buf = self.alloc(max(n * _lib.WINDOW, 1))
        try:
            self.synth_windows_dev(first, n, buf.ptr, seed)
            self.sync()
            return buf.download((n, _lib.WINDOW), np.uint8)
        finally:
            buf.free()

    # -- measurement --------------------------------------------------------------------
    def profile_enable(self, on=True):
        check(self.lib.gnn_profile_enable(self.ctx, 1 if on else 0))

    def profile_reset(self):
        check(self.lib.gnn_profile_reset(self.ctx))

    def profile_get(self, kernel_id: int):
        ms, n = C.c_double(), C.c_int64()
        check(self.lib.gnn_profile_get(self.ctx, kernel_id, C.byref(ms), C.byref(n)))
        return ms.value, n.value
