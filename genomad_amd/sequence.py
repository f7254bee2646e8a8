"""Host-side FASTA -> padded 6 kbp windows, mirroring the reference's data preparation.

Mirrors (same rules, same order, same error behaviour):
  genomad/sequence.py:96-121   read_fasta(filepath, strip_n=True)
  genomad/sequence.py:124-131  check_fasta
  genomad/sequence.py:150-167  seq_windows(seq, 6000, 2500, max_windows)
  genomad/modules/nn_classification.py:54-82  generate_data (window filter, upper-casing, ljust)
without the per-window Python objects and the TFRecord round trip: a contig is turned into rows of
one (n_windows, 6000) uint8 array with numpy slicing.  Tokenising and everything after it happens
on the GPU (libgenomad_nn_hip.so).
"""
import bz2
import gzip
import lzma
import sys
from pathlib import Path
from typing import Iterator, List, Tuple

import numpy as np

WINDOW = 6000
MIN_TAIL = 2500
MAX_N = 4000


def compression_of(path) -> str:
    """Magic-byte sniffing, genomad/utils.py:126-149: 'gzip' | 'bzip2' | 'xz' | 'zstd' | 'uncompressed'."""
    with open(path, "rb") as fin:
        sig = fin.read(8)
    if sig[:2] == b"\x1f\x8b":
        return "gzip"
    if sig[:3] == b"\x42\x5a\x68":
        return "bzip2"
    if sig[:7] == b"\xfd\x37\x7a\x58\x5a\x00\x00":
        return "xz"
    if sig[:4] == b"\x28\xb5\x2f\xfd":
        return "zstd"
    return "uncompressed"


def open_text(path):
    """genomad/utils.py:152-168 (zstd only on Python >= 3.14, like the reference)."""
    kind = compression_of(path)
    if kind == "gzip":
        return gzip.open(path, "rt")
    if kind == "bzip2":
        return bz2.open(path, "rt")
    if kind == "xz":
        return lzma.open(path, "rt")
    if kind == "zstd" and sys.version_info >= (3, 14):
        from compression import zstd  # type: ignore
        return zstd.open(path, "rt")
    return open(path, "r")


def read_fasta(path, strip_n: bool = False) -> Iterator[Tuple[str, str]]:
    """Yield (header, sequence).  Text before the first '>' is skipped, only the trailing newline
    of every line is removed, ``strip_n`` strips leading/trailing n/N, empty records are dropped."""
    with open_text(path) as fin:
        header, chunks = None, []

        def flush():
            s = "".join(chunks)
            if strip_n:
                s = s.strip("nN")
            return s

        for line in fin:
            if line[0] == ">":
                if header is not None:
                    s = flush()
                    if len(s):
                        yield header, s
                header, chunks = line.removesuffix("\n")[1:], []
            elif header is not None:
                chunks.append(line.removesuffix("\n"))
        if header is not None:
            s = flush()
            if len(s):
                yield header, s


def accession(header: str) -> str:
    return header.split()[0]   # genomad/sequence.py:24-25


def check_fasta_py(path) -> bool:
    """False if the file has no record or two records share an accession (sequence.py:124-131)."""
    acc = [accession(h) for h, _ in read_fasta(path)]
    return bool(acc) and len(acc) == len(set(acc))


def check_fasta(path, chunk_bytes: int = 256 << 20) -> bool:
    """:func:`check_fasta_py` with the line loop done by the library's packer in index mode, over
    record-aligned chunks of the (decompressed) text: host memory is bounded by one chunk plus the set of
    accessions, like the reference's line-by-line read (sequence.py:124-131)."""
    seen, n = set(), 0
    for chunk in iter_text_chunks(path, chunk_bytes):
        acc, _, _ = _pack(chunk, strip_n=False, copy=False)
        n += len(acc)
        seen.update(acc)
        if len(seen) != n:
            return False
    return n > 0


def _decode_header(raw: bytes) -> str:
    """Header bytes -> str.  The reference reads the file in text mode with the locale's encoding (UTF-8 in
    practice, utils.py:152-168), so non-ASCII header text is legal; ``surrogateescape`` keeps undecodable
    bytes round-trippable instead of raising on one rank of a multi-rank run."""
    return raw.decode("utf-8", "surrogateescape")


def window_spans(length: int, single_window: bool = False) -> List[Tuple[int, int]]:
    """seq_windows(seq, 6000, 2500, max_windows=1 if single_window else None) as (start, end)."""
    spans, win = [], 0
    while win * WINDOW < length:
        a, b = win * WINDOW, min((win + 1) * WINDOW, length)
        if b - a < MIN_TAIL:
            if win == 0:
                spans.append((a, b))
            break
        spans.append((a, b))
        win += 1
        if single_window and win == 1:
            break
    return spans


def contig_windows(seq: str, single_window: bool = False) -> np.ndarray:
    """(n, 6000) uint8: the windows generate_data keeps for one contig, upper-cased, 'N'-padded.

    The skip rule counts literal upper-case 'N' on the RAW sequence (Sequence.count,
    sequence.py:38-39) and never applies to window 0 (nn_classification.py:70-71).
    """
    raw = np.frombuffer(seq.encode("ascii"), dtype=np.uint8)
    spans = window_spans(len(raw), single_window)
    keep = [(a, b) for i, (a, b) in enumerate(spans)
            if i == 0 or int(np.count_nonzero(raw[a:b] == 78)) <= MAX_N]
    out = np.full((len(keep), WINDOW), 78, dtype=np.uint8)
    if keep:
        up = np.frombuffer(seq.upper().encode("ascii"), dtype=np.uint8)
        for i, (a, b) in enumerate(keep):
            out[i, :b - a] = up[a:b]
    return out


def encode_fasta(path, single_window: bool = False):
    """(contig_names, contig_ids, windows) like generate_data (nn_classification.py:54-82): names are
    accessions of the records that survive strip_n, ids index into them, one id per kept window."""
    names, ids, wins = [], [], []
    for cid, (header, seq) in enumerate(read_fasta(path, strip_n=True)):
        names.append(accession(header))
        w = contig_windows(seq, single_window)
        wins.append(w)
        ids.extend([cid] * len(w))
    windows = np.concatenate(wins) if wins else np.zeros((0, WINDOW), dtype=np.uint8)
    return np.array(names), np.array(ids, dtype=np.int64), windows


def _read_text_bytes(path) -> bytes:
    """Whole file as bytes, decompressed (utils.py:126-171 magic-byte sniffing), newlines normalised
    the way the reference's text-mode read does (universal newlines)."""
    kind = compression_of(path)
    opener = {"gzip": gzip.open, "bzip2": bz2.open, "xz": lzma.open}.get(kind)
    if kind == "zstd" and sys.version_info >= (3, 14):
        from compression import zstd  # type: ignore
        opener = zstd.open
    with (opener(path, "rb") if opener else open(path, "rb")) as fin:
        data = fin.read()
    if b"\r" in data:
        data = data.replace(b"\r\n", b"\n").replace(b"\r", b"\n")
    return data


def read_fasta_packed_py(path, strip_n: bool = True):
    """Pure-Python/numpy form of :func:`read_fasta_packed` (the readable specification the native
    packer is tested against; ≈ 0.2 GB/s)."""
    data = _read_text_bytes(path)
    records = (b"\n" + data).split(b"\n>")[1:]            # text before the first header line is dropped
    names, chunks, lengths = [], [], []
    for rec in records:
        header, _, body = rec.partition(b"\n")
        s = body.replace(b"\n", b"")
        if strip_n:
            s = s.strip(b"nN")
        if len(s):
            names.append(accession(_decode_header(header)))
            chunks.append(s)
            lengths.append(len(s))
    offsets = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(np.asarray(lengths, dtype=np.int64), out=offsets[1:])
    seq = np.frombuffer(b"".join(chunks), dtype=np.uint8) if chunks else np.zeros(0, dtype=np.uint8)
    return np.array(names), seq, offsets


def record_aligned_range(path, rank: int, world: int, piece: int = 0, pieces: int = 1) -> Tuple[int, int]:
    """Half-open byte range of ``rank``'s share of an UNCOMPRESSED FASTA file, aligned to record
    starts: both ends are moved forward to the next line that begins with '>' (a record belongs to
    the rank whose nominal range contains its '>').  Ranges of consecutive ranks tile the file, so
    every rank can read and pack only its own part (contigs shard embarrassingly).  Rank 0 starts
    at 0: text before the first header is dropped by the packer anyway.  ``piece``/``pieces`` cut a
    rank's share further the same way (used to pack piece k+1 while the GPU classifies piece k)."""
    if world < 1 or not (0 <= rank < world) or pieces < 1 or not (0 <= piece < pieces):
        raise ValueError("bad shard arguments")
    size = Path(path).stat().st_size
    rank, world = rank * pieces + piece, world * pieces

    def align(a: int) -> int:
        if a <= 0:
            return 0
        if a >= size:
            return size
        with open(path, "rb") as fin:
            pos = a - 1                                   # "\n>" may straddle the nominal boundary
            while pos < size:
                fin.seek(pos)
                block = fin.read(1 << 20)
                if not block:
                    break
                k = block.find(b"\n>")
                if k >= 0:
                    return pos + k + 1
                if len(block) < 2:                        # the last byte cannot start a "\n>" pair
                    break
                pos += len(block) - 1                     # keep one byte of overlap
        return size

    return align(size * rank // world), align(size * (rank + 1) // world)


def _read_text_array(path, byte_range=None) -> np.ndarray:
    """Writable uint8 array with the decompressed file contents (no newline normalisation);
    ``byte_range`` (uncompressed files only) reads just [start, end)."""
    kind = compression_of(path)
    opener = {"gzip": gzip.open, "bzip2": bz2.open, "xz": lzma.open}.get(kind)
    if kind == "zstd" and sys.version_info >= (3, 14):
        from compression import zstd  # type: ignore
        opener = zstd.open
    if opener is None:
        start, end = byte_range if byte_range is not None else (0, Path(path).stat().st_size)
        size = max(end - start, 0)
        buf = bytearray(size)
        with open(path, "rb", buffering=0) as fin:
            fin.seek(start)
            got, view = 0, memoryview(buf)
            while got < size:
                k = fin.readinto(view[got:])
                if not k:
                    break
                got += k
        return np.frombuffer(buf, dtype=np.uint8)[:got]
    if byte_range is not None:
        raise ValueError("byte ranges need an uncompressed file")
    with opener(path, "rb") as fin:
        return np.frombuffer(bytearray(fin.read()), dtype=np.uint8)


def _open_binary(path):
    kind = compression_of(path)
    opener = {"gzip": gzip.open, "bzip2": bz2.open, "xz": lzma.open}.get(kind)
    if kind == "zstd" and sys.version_info >= (3, 14):
        from compression import zstd  # type: ignore
        opener = zstd.open
    return opener(path, "rb") if opener else open(path, "rb")


def iter_text_chunks(path, chunk_bytes: int = 128 << 20):
    """The decompressed file as writable uint8 arrays of about ``chunk_bytes``, each cut at a record start
    (a line beginning with '>'), so that every chunk can be packed on its own and the chunks tile the
    file.  A record longer than a chunk simply makes that chunk longer."""
    with _open_binary(path) as fin:
        carry = []                       # blocks of a record that is longer than one read: joined ONCE, when it ends
        while True:
            block = fin.read(chunk_bytes)
            if not block:
                break
            # the record start may straddle two reads ("\n" ends one block, ">" starts the next)
            cut = block.rfind(b"\n>")
            if cut < 0 and carry and carry[-1].endswith(b"\n") and block.startswith(b">"):
                yield np.frombuffer(bytearray(b"".join(carry)), dtype=np.uint8)
                carry = [block]
                continue
            if cut < 0:
                carry.append(block)
                continue
            carry.append(block[:cut + 1])
            yield np.frombuffer(bytearray(b"".join(carry)), dtype=np.uint8)
            carry = [block[cut + 1:]]
        if carry and any(carry):
            yield np.frombuffer(bytearray(b"".join(carry)), dtype=np.uint8)


def index_accessions(text: np.ndarray):
    """Accessions of ALL records of a text array (no N stripping: what check_fasta counts), without consuming it."""
    acc, _, _ = _pack(text, strip_n=False, copy=False)
    return acc


def _digest_of_accession(acc: str) -> int:
    """Python mirror of gnn_fasta_accession_digests' hash (FNV-1a 64 of the accession's bytes + the splitmix64 mixer)."""
    h, m = 0xcbf29ce484222325, (1 << 64) - 1
    for b in acc.encode("utf-8", "surrogateescape"):
        h = ((h ^ b) * 0x100000001b3) & m
    h ^= h >> 30
    h = (h * 0xbf58476d1ce4e5b9) & m
    h ^= h >> 27
    h = (h * 0x94d049bb133111eb) & m
    return h ^ (h >> 31)


def accession_digests_of_text(text: np.ndarray) -> np.ndarray:
    """uint64 digests of the accessions of ALL records of a text array with a non-empty raw sequence (what check_fasta counts,
    genomad/sequence.py:124-131), without consuming it: one native pass (``gnn_fasta_accession_digests``).  A text with a
    non-ASCII byte inside a first header token (Python's ``split()`` knows non-ASCII white space) takes the Python route -
    the same digests, computed from :func:`index_accessions`."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    nh, hb, cr = C.c_int64(), C.c_int64(), C.c_int()
    _lib.check(lib.gnn_fasta_scan(text.ctypes.data, len(text), C.byref(nh), C.byref(hb), C.byref(cr)))
    if not cr.value:
        out = np.empty(max(nh.value, 1), dtype="<u8")
        nrec, odd = C.c_int64(), C.c_int()
        _lib.check(lib.gnn_fasta_accession_digests(text.ctypes.data, len(text), out.ctypes.data, nh.value, C.byref(nrec), C.byref(odd)))
        if not odd.value:
            return out[:nrec.value].copy()
    return np.array([_digest_of_accession(a) for a in index_accessions(text)], dtype="<u8")


def pack_text(text: np.ndarray, strip_n: bool = True):
    """(names, seq, offsets) of the records in a writable text array (consumed: packed in place)."""
    names, seq, offsets = _pack(text, strip_n)
    return (np.array(names) if names else np.zeros(0, dtype="<U1")), seq, offsets


def _pack(text: np.ndarray, strip_n: bool, copy: bool = True):
    """Run gnn_fasta_scan / gnn_fasta_pack over a writable text array.  copy=True packs IN PLACE
    (``text`` is consumed) and returns (names, seq view, offsets); copy=False is the index mode."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    nh, hb, cr = C.c_int64(), C.c_int64(), C.c_int()
    _lib.check(lib.gnn_fasta_scan(text.ctypes.data, len(text), C.byref(nh), C.byref(hb), C.byref(cr)))
    if cr.value:        # universal newlines, as the reference's text-mode read (utils.py:152-168); rare
        text = np.frombuffer(bytearray(text.tobytes().replace(b"\r\n", b"\n").replace(b"\r", b"\n")), dtype=np.uint8)
        _lib.check(lib.gnn_fasta_scan(text.ctypes.data, len(text), C.byref(nh), C.byref(hb), C.byref(cr)))
    cap = nh.value
    offsets = np.zeros(cap + 1, dtype=np.int64)
    hoff = np.zeros(cap + 1, dtype=np.int64)
    headers = np.empty(max(hb.value, 1), dtype=np.uint8)
    nrec = C.c_int64()
    _lib.check(lib.gnn_fasta_pack(text.ctypes.data, len(text), int(bool(strip_n)), text.ctypes.data if copy else None,
                                  offsets.ctypes.data, headers.ctypes.data, hoff.ctypes.data, cap, C.byref(nrec)))
    k = nrec.value
    offsets = offsets[:k + 1].copy()
    hraw = headers.tobytes()
    names = [accession(_decode_header(hraw[hoff[i]:hoff[i + 1]])) for i in range(k)]
    return names, (text[:offsets[-1]] if copy else None), offsets


def read_fasta_packed(path, strip_n: bool = True, byte_range=None):
    """(names, seq, offsets): every record that ``read_fasta(path, strip_n)`` yields, packed into ONE
    uint8 buffer of raw (case-preserved) sequence bytes; contig i is seq[offsets[i]:offsets[i+1]].

    Same record rules as :func:`read_fasta` (header = a line starting with '>', only '\\n' is
    removed, leading/trailing n/N stripped, empty records dropped); the line work is done by the
    library's host-side packer (``gnn_fasta_pack``: memchr/memmove, IN PLACE in the buffer the file
    was read into, several GB/s) so that real inputs keep up with the device.  ``byte_range`` (from
    :func:`record_aligned_range`) restricts the read to one rank's share of an uncompressed file.
    """
    names, seq, offsets = _pack(_read_text_array(path, byte_range), strip_n)
    return (np.array(names) if names else np.zeros(0, dtype="<U1")), seq, offsets


def candidate_spans(offsets: np.ndarray, single_window: bool = False):
    """Vectorised seq_windows(seq, 6000, 2500, max_windows) over all contigs (sequence.py:150-167):
    returns (starts int64, lens int32, contig_ids int64, window_n int32) of every candidate window,
    before the N-content rule."""
    lengths = np.diff(offsets)
    nfull, rem = lengths // WINDOW, lengths % WINDOW
    nwin = nfull + (rem >= MIN_TAIL)
    nwin = np.where((lengths > 0) & (nwin == 0), 1, nwin)          # window 0 is always yielded
    if single_window:
        nwin = np.minimum(nwin, 1)
    ids = np.repeat(np.arange(len(lengths), dtype=np.int64), nwin)
    first = np.cumsum(nwin) - nwin
    window_n = (np.arange(int(nwin.sum()), dtype=np.int64) - np.repeat(first, nwin)).astype(np.int32)
    starts = offsets[:-1][ids] + window_n.astype(np.int64) * WINDOW
    lens = np.minimum(WINDOW, offsets[1:][ids] - starts).astype(np.int32)
    return starts, lens, ids, window_n


def prefix_of(input_path: Path) -> str:
    """nn_classification.py:106-108: stem, minus one more extension if the file is compressed."""
    prefix = Path(input_path).stem
    if compression_of(input_path) != "uncompressed":
        prefix = prefix.rsplit(".", 1)[0]
    return prefix
