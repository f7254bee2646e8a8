// Host-side FASTA record packer (no device code): the record rules of the reference's
// read_fasta(strip_n) (genomad/sequence.py:96-121) applied to an in-memory text buffer with
// memchr/memcpy instead of a Python line loop, so that reading real inputs keeps up with the
// kernels (one MI355X consumes ≈ 0.83 GB/s of sequence; the pure-Python reader does 0.2 GB/s).
//
// Rules (identical to genomad_amd.sequence.read_fasta, which stays the readable specification and
// is what the tests compare this against):
//   * a header is a line whose FIRST byte is '>'; everything before the first header is dropped;
//   * only the '\n' ending a line is removed — other whitespace inside a line is kept;
//   * the record's sequence is the concatenation of its lines; with strip_n its leading and trailing
//     'n'/'N' bytes are removed; records that end up empty are dropped;
//   * universal newlines ('\r\n', '\r') are the caller's business (normalised before the call).
#include <cstring>

#include "gnn_common.h"

using namespace gnn;

extern "C" int gnn_fasta_scan(const uint8_t* text, int64_t n, int64_t* n_headers, int64_t* header_bytes, int* has_cr) {
    if ((!text && n > 0) || n < 0 || !n_headers || !header_bytes || !has_cr) {
        set_error("bad argument to gnn_fasta_scan");
        return GNN_ERR_ARG;
    }
    int64_t count = 0, bytes = 0, p = 0;
    while (p < n) {
        const uint8_t* q = static_cast<const uint8_t*>(memchr(text + p, '>', (size_t)(n - p)));
        if (!q) break;
        const int64_t i = q - text;
        p = i + 1;
        if (i == 0 || text[i - 1] == '\n') {
            const uint8_t* e = static_cast<const uint8_t*>(memchr(text + p, '\n', (size_t)(n - p)));
            const int64_t le = e ? e - text : n;
            ++count;
            bytes += le - p;
            p = le;
        }
    }
    *n_headers = count;
    *header_bytes = bytes;
    *has_cr = n > 0 && memchr(text, '\r', (size_t)n) != nullptr;
    return GNN_OK;
}

extern "C" int gnn_fasta_pack(const uint8_t* text, int64_t n, int strip_n, uint8_t* seq_out, int64_t* offsets,
                              uint8_t* headers_out, int64_t* header_offsets, int64_t capacity, int64_t* n_records) {
    // seq_out == NULL: index mode — nothing is copied, offsets hold cumulative raw lengths (what
    // check_fasta needs: the headers of the records whose sequence is non-empty); no stripping then
    const bool copy = seq_out != nullptr;
    if ((!text && n > 0) || n < 0 || !offsets || !header_offsets || !n_records || capacity < 0 || (!copy && strip_n)) {
        set_error("bad argument to gnn_fasta_pack");
        return GNN_ERR_ARG;
    }
    auto is_n = [](uint8_t b) { return b == 'n' || b == 'N'; };
    // out <= p at all times, so packing in place (seq_out == text) only ever moves bytes towards lower
    // addresses; header text is copied out before the record's sequence can overwrite it
    int64_t p = 0, out = 0, nrec = 0, rec_begin = 0, hout = 0;
    bool in_rec = false, overflow = false;
    offsets[0] = 0;
    header_offsets[0] = 0;
    auto finish = [&]() {
        if (!in_rec) return;
        int64_t b = rec_begin, e = out;
        if (strip_n) {
            while (b < e && is_n(seq_out[b])) ++b;
            while (e > b && is_n(seq_out[e - 1])) --e;
            if (b > rec_begin) memmove(seq_out + rec_begin, seq_out + b, (size_t)(e - b));
            e = rec_begin + (e - b);
        }
        if (e > rec_begin) {
            offsets[++nrec] = e;
            header_offsets[nrec] = hout;
            out = e;
        } else {
            out = rec_begin;                     // dropped: its header text is discarded too
            hout = header_offsets[nrec];
        }
    };
    while (p < n) {
        const uint8_t* q = static_cast<const uint8_t*>(memchr(text + p, '\n', (size_t)(n - p)));
        const int64_t le = q ? q - text : n;             // line = [p, le)
        if (text[p] == '>') {                            // (an empty line has text[p] == '\n')
            finish();
            if (nrec >= capacity) {
                overflow = true;
                break;
            }
            in_rec = true;
            rec_begin = out;
            if (le > p + 1) memcpy(headers_out + hout, text + p + 1, (size_t)(le - p - 1));
            hout += le - p - 1;
        } else if (in_rec && le > p) {
            if (copy) memmove(seq_out + out, text + p, (size_t)(le - p));
            out += le - p;
        }
        p = le + 1;
    }
    if (overflow) {
        set_error("gnn_fasta_pack: more records than capacity (size it with gnn_fasta_scan)");
        return GNN_ERR_ARG;
    }
    finish();
    *n_records = nrec;
    return GNN_OK;
}

// 64-bit digests of the ACCESSIONS (header.split()[0], genomad/sequence.py:24-25) of every record with a non-empty raw sequence -
// what the reference's check_fasta counts (sequence.py:124-131) - in ONE pass over the text, for the sharded validation of
// genomad_amd/sharding.py (the Python side used to scan the text twice and decode + hash every header in a loop).
// The accession is the first run of bytes outside Python's ASCII whitespace (\t \n \v \f \r, 0x1c-0x1f, space).  Python's
// str.split() also splits at non-ASCII white space (U+0085, U+00A0, U+2028 ...), all of which are encoded with bytes >= 0x80: a
// header with such a byte before the end of its first token - or an empty accession, on which the reference raises - sets
// *needs_python, and the caller recomputes THAT piece with the Python mirror of this hash (genomad_amd/sequence.py).
// digest = FNV-1a 64 of the accession bytes, finished with the splitmix64 mixer.
static inline bool py_ascii_space(uint8_t b) { return (b >= 0x09 && b <= 0x0d) || (b >= 0x1c && b <= 0x20); }

extern "C" int gnn_fasta_accession_digests(const uint8_t* text, int64_t n, uint64_t* digests, int64_t capacity, int64_t* n_records,
                                           int* needs_python) {
    if ((!text && n > 0) || n < 0 || (!digests && capacity > 0) || capacity < 0 || !n_records || !needs_python) {
        set_error("bad argument to gnn_fasta_accession_digests");
        return GNN_ERR_ARG;
    }
    int64_t p = 0, nrec = 0, seq_len = 0;
    bool in_rec = false, odd = false;
    uint64_t cur = 0;
    auto finish = [&]() -> bool {
        if (in_rec && seq_len > 0) {
            if (nrec >= capacity) return false;
            digests[nrec++] = cur;
        }
        return true;
    };
    while (p < n) {
        const uint8_t* q = static_cast<const uint8_t*>(memchr(text + p, '\n', (size_t)(n - p)));
        const int64_t le = q ? q - text : n;
        if (text[p] == '>') {
            if (!finish()) {
                set_error("gnn_fasta_accession_digests: more records than capacity (size it with gnn_fasta_scan)");
                return GNN_ERR_ARG;
            }
            in_rec = true;
            seq_len = 0;
            int64_t a = p + 1;
            while (a < le && py_ascii_space(text[a])) ++a;
            int64_t b = a;
            uint64_t h = 0xcbf29ce484222325ull;
            while (b < le && !py_ascii_space(text[b])) {
                odd |= text[b] >= 0x80;
                h = (h ^ text[b]) * 0x100000001b3ull;
                ++b;
            }
            odd |= b == a;
            h ^= h >> 30;
            h *= 0xbf58476d1ce4e5b9ull;
            h ^= h >> 27;
            h *= 0x94d049bb133111ebull;
            h ^= h >> 31;
            cur = h;
        } else if (in_rec) {
            seq_len += le - p;
        }
        p = le + 1;
    }
    if (!finish()) {
        set_error("gnn_fasta_accession_digests: more records than capacity (size it with gnn_fasta_scan)");
        return GNN_ERR_ARG;
    }
    *n_records = nrec;
    *needs_python = odd ? 1 : 0;
    return GNN_OK;
}
