"""How the FASTA validation scales when it is sharded with the contigs: the sequential whole-file
check_fasta that rank 0 ran until round 3 against what every rank of a G-rank main() now does for ITS byte range (index pass
over the records + 64-bit digests of the accessions; genomad_amd/sharding.fasta_verdict gathers them).  CPU only.

    python scripts/validation_shard_bench.py [MB]      (default 512)
"""
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomad_amd import sequence, sharding  # noqa: E402


def _share(args):
    path, rank, world = args
    t = time.perf_counter()
    pieces = 4
    n = 0
    for k in range(pieces):
        text = sequence._read_text_array(path, sequence.record_aligned_range(path, rank, world, k, pieces))
        acc = sequence.index_accessions(text)
        n += len(sharding.accession_digests(acc))
    return time.perf_counter() - t, n


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    rng = np.random.default_rng(0)
    path = os.path.join(tempfile.mkdtemp(prefix="valbench_"), "meta.fna")
    with open(path, "wb") as f:
        i, done = 0, 0
        line = rng.choice(np.frombuffer(b"ACGT", np.uint8), 1 << 20).tobytes()
        while done < mb << 20:
            L = int(np.exp(rng.uniform(np.log(1e3), np.log(5e5))))
            f.write(b">contig_%d metagenome\n" % i)
            for a in range(0, L, 80):
                off = int(rng.integers(0, (1 << 20) - 80))
                f.write(line[off:off + min(80, L - a)] + b"\n")
            i, done = i + 1, done + L
    size = os.path.getsize(path)
    sequence.check_fasta(path)                                  # page cache warm
    t = time.perf_counter()
    ok = sequence.check_fasta(path)
    seq_s = time.perf_counter() - t
    print(f"{size / 1e6:.0f} MB FASTA, {i} records; sequential check_fasta (rank 0, rounds 1-3): {seq_s:.3f} s = {size / seq_s / 1e9:.2f} GB/s, verdict {ok}")
    for world in (1, 2, 4, 8):
        with mp.get_context("spawn").Pool(world) as pool:
            pool.map(_share, [(path, r, world) for r in range(world)])          # warm the workers
            res = pool.map(_share, [(path, r, world) for r in range(world)])
        worst = max(r[0] for r in res)
        print(f"  {world} rank(s): slowest rank's share {worst:.3f} s ({sum(r[1] for r in res)} records) -> {seq_s / worst:.2f}x the sequential check")
    os.remove(path)


if __name__ == "__main__":
    main()
