"""TFRecord wire compatibility for ``<prefix>_encoded_sequences/*.tfrec`` — SURVEY.md §8f rank 4.

The reference stores every tokenised window as ``tf.train.Example{features{feature{"sequence":
Feature{int64_list{value: 5997 ints}}}}}`` in TFRecord files of 10 000 records named
``<window_count>.tfrec`` (genomad/modules/nn_classification.py:43-52, :75-81) and reads them back with
``tf.io.parse_single_example`` (:87-91).  This module writes and reads that format without
TensorFlow, so that a reference run can resume from a directory encoded here and vice versa.

Wire format (public TFRecord / protobuf encodings):
  record   = uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data)
  masked   = rotr15(crc) + 0xa282ead8 (mod 2**32), crc = CRC-32C (Castagnoli)
  Example  = 0A len( Features = 0A len( MapEntry = 0A 08 "sequence" 12 len( Feature = 1A len(
             Int64List = 0A len( packed varints ))))))
TensorFlow is not installed here, so byte identity with TF's own writer is by construction from the
published formats, not by comparison ("unverifiable here", like the rest of the TF half).
"""
import struct
from pathlib import Path

import numpy as np

from . import _lib

RECORDS_PER_FILE = 10_000     # nn_classification.py:58
N_TOKENS = 5997
_KEY = b"sequence"


def _crc(data: bytes) -> int:
    lib = _lib.load()
    return lib.gnn_crc32c(data, len(data))


def _masked(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf: bytes, pos: int):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def encode_example(tokens: np.ndarray) -> bytes:
    """Serialised tf.train.Example for one window's tokens (values in [0, 256])."""
    t = np.asarray(tokens, dtype=np.uint16)
    big = t >= 128
    packed = np.empty(len(t) + int(big.sum()), dtype=np.uint8)          # varints: 1 byte, or 2 if >= 128
    idx = np.arange(len(t)) + np.concatenate(([0], np.cumsum(big)[:-1]))
    packed[idx] = np.where(big, (t & 0x7F) | 0x80, t).astype(np.uint8)
    packed[idx[big] + 1] = (t[big] >> 7).astype(np.uint8)
    int64_list = b"\x0a" + _varint(len(packed)) + packed.tobytes()
    feature = b"\x1a" + _varint(len(int64_list)) + int64_list
    entry = b"\x0a" + _varint(len(_KEY)) + _KEY + b"\x12" + _varint(len(feature)) + feature
    features = b"\x0a" + _varint(len(entry)) + entry
    return b"\x0a" + _varint(len(features)) + features


def decode_example(data: bytes) -> np.ndarray:
    """Inverse of :func:`encode_example` (also accepts un-packed int64 lists)."""
    def field(buf, pos, want_tag):
        tag, pos = _read_varint(buf, pos)
        if tag != want_tag:
            raise ValueError(f"unexpected protobuf tag {tag:#x}, wanted {want_tag:#x}")
        n, pos = _read_varint(buf, pos)
        return buf[pos:pos + n], pos + n
    features, _ = field(data, 0, 0x0A)
    entry, _ = field(features, 0, 0x0A)
    key, pos = field(entry, 0, 0x0A)
    if key != _KEY:
        raise ValueError(f"feature key {key!r}, expected b'sequence'")
    feature, _ = field(entry, pos, 0x12)
    int64_list, _ = field(feature, 0, 0x1A)
    out, pos = [], 0
    while pos < len(int64_list):
        tag, pos = _read_varint(int64_list, pos)
        if tag == 0x0A:                                   # packed
            n, pos = _read_varint(int64_list, pos)
            end = pos + n
            while pos < end:
                v, pos = _read_varint(int64_list, pos)
                out.append(v)
        elif tag == 0x08:                                 # one un-packed value
            v, pos = _read_varint(int64_list, pos)
            out.append(v)
        else:
            raise ValueError(f"unexpected tag {tag:#x} in Int64List")
    return np.array(out, dtype=np.int64)


def write_file(path, token_rows) -> None:
    with open(path, "wb") as f:
        for row in token_rows:
            data = encode_example(row)
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", _masked(_crc(head))) + data + struct.pack("<I", _masked(_crc(data))))


def read_file(path) -> np.ndarray:
    rows = []
    with open(path, "rb") as f:
        while True:
            head = f.read(8)
            if not head:
                break
            (n,) = struct.unpack("<Q", head)
            (hcrc,) = struct.unpack("<I", f.read(4))
            if hcrc != _masked(_crc(head)):
                raise ValueError(f"{path}: corrupt record length")
            data = f.read(n)
            (dcrc,) = struct.unpack("<I", f.read(4))
            if dcrc != _masked(_crc(data)):
                raise ValueError(f"{path}: corrupt record data")
            rows.append(decode_example(data))
    return np.stack(rows) if rows else np.zeros((0, N_TOKENS), dtype=np.int64)


def write_dir(directory, tokens: np.ndarray) -> list:
    """Split (n, 5997) tokens into files exactly like generate_data does (nn_classification.py:75-81):
    a file is flushed whenever the running window count hits a multiple of 10 000 and is named after
    that count; the remainder goes to ``<n>.tfrec``."""
    directory = Path(directory)
    written = []
    for a in range(0, len(tokens), RECORDS_PER_FILE):
        b = min(a + RECORDS_PER_FILE, len(tokens))
        p = directory / f"{b}.tfrec"
        write_file(p, tokens[a:b])
        written.append(p)
    return written


def read_dir(directory) -> np.ndarray:
    """All records of ``*.tfrec`` in natural (numeric) file order, as utils.natsort does (:294-296)."""
    files = sorted(Path(directory).glob("*.tfrec"), key=lambda p: int(p.stem) if p.stem.isdigit() else 1 << 62)
    parts = [read_file(p) for p in files]
    return np.concatenate(parts) if parts else np.zeros((0, N_TOKENS), dtype=np.int64)


def tokens_to_bases(tokens: np.ndarray) -> np.ndarray:
    """(n, 5997) tokens -> (n, 6000) uint8 windows that tokenise back to exactly these tokens.

    Base i is covered by the 4-mers starting at i-3 .. i; any non-zero covering token determines it,
    and a base covered only by zero tokens is written as 'N' (every choice of which base carried the
    non-ACGT byte gives the same tokens, hence the same scores).  Lets a directory encoded by the
    reference be classified here.
    """
    t = np.asarray(tokens, dtype=np.int64)
    n, L = t.shape
    out = np.full((n, L + 3), ord("N"), dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for j in range(4):                                   # base i = j-th base of the 4-mer starting at i-j
        code = ((t - 1) >> (2 * (3 - j))) & 3
        valid = t > 0
        view = out[:, j:j + L]
        view[valid] = acgt[code[valid]]
    return out
