"""CPU tests of the host side: FASTA/windowing mirror vs the reference goldens, the drop-in
main() file contract (JSON / NPZ / TSV / resume / errors) with a fake scoring backend, sharding
logic and the world_size-2 gloo gather."""
import gzip
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from genomad_amd import nn_classification as nnc
from genomad_amd import sequence, sharding
from oracle import sequence_oracle


class FakeBackend:
    """Deterministic stand-in for the GPU (tests only): score = f(window bytes)."""

    def score(self, windows):
        w = np.asarray(windows, dtype=np.float64)
        s = np.stack([(w == c).mean(axis=1) for c in (65, 67, 71)], axis=1) + 0.01
        return (s / s.sum(axis=1, keepdims=True)).astype(np.float32)

    def segment_mean(self, scores, ids, n):
        out = sequence_oracle.segment_mean(np.asarray(scores, np.float32), np.asarray(ids))
        return np.concatenate([out, np.zeros((n - len(out), 3), np.float32)])


def test_encode_fasta_matches_reference_golden(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "fasta_golden.json")))
    path = os.path.join(golden_dir, "fasta_fixture.fna.gz")
    assert sequence.check_fasta(path) == g["check_fasta"]
    assert sequence.prefix_of(path) == "fasta_fixture"
    for key, single in (("all", False), ("single", True)):
        names, ids, wins = sequence.encode_fasta(path, single)
        assert list(names) == g[key]["contig_names"] and list(ids) == g[key]["contig_ids"]
        toks = sequence_oracle.tokenize_closed_form(wins).astype("<u2")
        dig = [hashlib.sha256(w.tobytes()).hexdigest()[:16] + ":" + hashlib.sha256(t.tobytes()).hexdigest()[:16]
               for w, t in zip(wins, toks)]
        assert dig == g[key]["window_digests"]


def test_window_spans_match_reference_golden(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "windowing_golden.json")))["window_lengths"]
    for L, want in g.items():
        assert [b - a for a, b in sequence.window_spans(int(L))] == want["all"]
        assert [b - a for a, b in sequence.window_spans(int(L), True)] == want["single"]


def _write_fasta(path, recs, gz=False):
    text = "".join(f">{n}\n{s}\n" for n, s in recs)
    if gz:
        with gzip.open(path, "wt") as f:
            f.write(text)
    else:
        path.write_text(text)


def test_main_file_contract_and_resume(tmp_path):
    rng = np.random.default_rng(5)
    recs = [("c1 some description", "".join(rng.choice(list("ACGT"), 13000))),
            ("c2", "NN" + "".join(rng.choice(list("ACGT"), 2600)) + "nn"),
            ("c3", "N" * 50)]
    fa = tmp_path / "sample.fna.gz"
    _write_fasta(fa, recs, gz=True)
    out = tmp_path / "out"
    nnc.main(fa, out, False, 128, False, 1, False, False, _backend=FakeBackend())
    d = out / "sample_nn_classification"
    info = json.loads((d / "sample_nn_classification.json").read_text())
    assert list(info) == ["module", "input", "input_md5", "start_time", "parameters"]
    assert info["module"] == "nn_classification" and info["input"] == "sample.fna.gz"
    assert info["parameters"] == {"single_window": False}
    assert info["input_md5"] == hashlib.md5(fa.read_bytes()).hexdigest()
    assert (d / "sample_nn_classification.json").read_text().endswith("}\n")
    z = np.load(d / "sample_nn_classification.npz")
    assert list(z["contig_names"]) == ["c1", "c2"]                 # the all-N record is dropped
    assert z["predictions"].shape == (2, 3) and z["predictions"].dtype == np.float32
    wid = np.load(d / "sample_encoded_sequences" / "sample_seq_window_id.npz")
    assert list(wid["contig_ids"]) == [0, 0, 1]                    # 13000 -> 6000+6000(+1000 dropped); 2600 -> 1
    lines = (d / "sample_nn_classification.tsv").read_text().splitlines()
    assert lines[0] == "seq_name\tchromosome_score\tplasmid_score\tvirus_score"
    name, *vals = lines[1].split("\t")
    assert name == "c1" and len(vals) == 3 and all(len(v.split(".")[1]) == 4 for v in vals)
    assert vals == [f"{x:.4f}" for x in z["predictions"][0]]
    assert (out / "sample_nn_classification.log").exists()
    # per-contig mean of window scores
    _, ids, wins = sequence.encode_fasta(fa)
    fb = FakeBackend()
    assert np.allclose(z["predictions"], fb.segment_mean(fb.score(wins), ids, 2))

    # resume: same input+params -> classification skipped (a backend that must not be called)
    class Boom:
        def score(self, w):
            raise AssertionError("classification should have been skipped")
        segment_mean = FakeBackend.segment_mean
    nnc.main(fa, out, False, 128, False, 1, False, False, _backend=Boom())
    # changed parameter -> recomputed; --cleanup removes the encoded directory
    nnc.main(fa, out, True, 128, False, 1, False, True, _backend=FakeBackend())
    assert not (d / "sample_encoded_sequences").exists()
    assert json.loads((d / "sample_nn_classification.json").read_text())["parameters"] == {"single_window": True}
    # restart forces recomputation even if nothing changed
    with pytest.raises(AssertionError, match="skipped"):
        nnc.main(fa, out, True, 128, True, 1, False, False, _backend=Boom())


def test_main_errors_exit_1(tmp_path, capsys):
    dup = tmp_path / "dup.fna"
    _write_fasta(dup, [("a", "ACGT" * 10), ("a", "ACGT" * 10)])
    with pytest.raises(SystemExit) as e:
        nnc.main(dup, tmp_path / "o1", False, 128, False, 1, False, False, _backend=FakeBackend())
    assert e.value.code == 1
    empty = tmp_path / "empty.fna"
    empty.write_text("")
    with pytest.raises(SystemExit) as e:
        nnc.main(empty, tmp_path / "o2", False, 128, False, 1, False, False, _backend=FakeBackend())
    assert e.value.code == 1
    alln = tmp_path / "alln.fna"                                   # passes check_fasta, yields no window
    _write_fasta(alln, [("a", "N" * 100)])
    with pytest.raises(SystemExit) as e:
        nnc.main(alln, tmp_path / "o3", False, 128, False, 1, False, False, _backend=FakeBackend())
    assert e.value.code == 1
    assert "No sequences were found" in capsys.readouterr().err


def test_provirus_pass_runs_when_find_proviruses_outputs_exist(tmp_path):
    fa = tmp_path / "g.fna"
    _write_fasta(fa, [("c1", "ACGT" * 2000)])
    out = tmp_path / "out"
    fp = out / "g_find_proviruses"
    fp.mkdir(parents=True)
    (fp / "g_find_proviruses.json").write_text(json.dumps({"input_md5": nnc.get_md5(fa), "module": "x", "parameters": {}}))
    (fp / "g_provirus.tsv").write_text("h\nc1|provirus_1_4000\n")
    _write_fasta(fp / "g_provirus.fna", [("c1|provirus_1_4000", "ACGT" * 1000)])
    (fp / "g_provirus_proteins.faa").write_text("")
    (fp / "g_provirus_genes.tsv").write_text("")
    nnc.main(fa, out, False, 128, False, 1, False, False, _backend=FakeBackend())
    z = np.load(out / "g_nn_classification" / "g_provirus_nn_classification.npz")
    assert list(z["provirus_names"]) == ["c1|provirus_1_4000"] and z["predictions"].shape == (1, 3)
    assert (out / "g_nn_classification" / "g_provirus_nn_classification.tsv").exists()


def test_product_path_fails_loudly_without_gpu_or_weights(tmp_path, monkeypatch):
    monkeypatch.delenv("GENOMAD_AMD_WEIGHTS", raising=False)
    monkeypatch.setattr(nnc, "_ENGINE", None)
    fa = tmp_path / "g.fna"
    _write_fasta(fa, [("c1", "ACGT" * 2000)])
    with pytest.raises((FileNotFoundError, RuntimeError)):
        nnc.main(fa, tmp_path / "out", False, 128, False, 1, False, False)


def test_shard_ranges_cover_and_order():
    for n in (0, 1, 7, 8, 9, 1000, 1_000_000):
        for g in (1, 2, 3, 8):
            r = [sharding.shard_range(n, g, k) for k in range(g)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(g - 1))
            assert sum(sharding.shard_counts(n, g)) == n
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def _gloo_worker(rank, world, port, n, q):
    from tests.gloo_comm import GlooComm
    comm = GlooComm(rank, world, port)
    rng = np.random.default_rng(0)
    windows = rng.integers(65, 85, (n, 6000), dtype=np.uint8)
    out = sharding.classify_sharded(windows, FakeBackend().score, comm)
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    comm.close()


@pytest.mark.parametrize("n", [11, 4])
def test_gloo_world_size_2_gather_equals_single_process(n):
    """N>1 path on CPU: two ranks shard the windows, rank 0 gathers; result must equal the
    single-process scores bit for bit (no cross-rank reduction is involved)."""
    mp = pytest.importorskip("torch.multiprocessing")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    windows = rng.integers(65, 85, (n, 6000), dtype=np.uint8)
    assert np.array_equal(got, FakeBackend().score(windows))


def _fake_contig_scores(seq, offsets):
    """Stand-in for the GPU (tests only): per-contig base composition + candidate window ids."""
    k = len(offsets) - 1
    out = np.zeros((k, 3), np.float32)
    for i in range(k):
        c = seq[offsets[i]:offsets[i + 1]]
        out[i] = [(c == b).mean() for b in (65, 67, 71)]
    return out, sequence.candidate_spans(offsets)[2]


def test_gather_contig_parts_single_process_orders_pieces(tmp_path):
    p = tmp_path / "meta.fna"
    _write_sharding_fasta(p)
    parts = []
    for k in (2, 0, 1):
        names, seq, offsets = sequence.read_fasta_packed(p, True, sequence.record_aligned_range(p, 0, 1, k, 3))
        parts.append((k, names, *_fake_contig_scores(seq, offsets)))
    names, preds, ids, total = sharding.gather_contig_parts(None, parts)
    n1, s1, o1 = sequence.read_fasta_packed(p)
    want_scores, want_ids = _fake_contig_scores(s1, o1)
    assert list(names) == list(n1) and np.array_equal(preds, want_scores)
    assert np.array_equal(ids, want_ids) and total == len(want_ids)
    with pytest.raises(ValueError, match="duplicate"):
        sharding.gather_contig_parts(None, [parts[0], parts[0]])


def _write_sharding_fasta(path):
    rng = np.random.default_rng(21)
    with open(path, "wb") as f:
        f.write(b"preamble that is not a record\n")
        for i in range(37):
            n = int(rng.integers(1, 30000))
            body = rng.choice(np.frombuffer(b"ACGTN", np.uint8), n).tobytes()
            if i % 7 == 3:
                body = b"N" * n                                   # dropped after strip_n
            f.write(b">ctg%d d=%d\n" % (i, n))
            w = int(rng.integers(40, 90))
            f.write(b"\n".join(body[j:j + w] for j in range(0, n, w)) + b"\n")
            if i % 5 == 0:
                f.write(b"\n")


def test_record_aligned_byte_ranges_tile_the_file(tmp_path):
    """Multi-rank front end: every rank packs only its record-aligned byte range; the concatenation
    over ranks equals packing the whole file, for any number of ranks (also more ranks than records)."""
    p = tmp_path / "meta.fna"
    _write_sharding_fasta(p)
    names, seq, offsets = sequence.read_fasta_packed(p)
    size = p.stat().st_size
    for world in (1, 2, 3, 5, 8, 64):
        ranges = [sequence.record_aligned_range(p, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == size
        assert all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
        parts = [sequence.read_fasta_packed(p, True, rg) for rg in ranges]
        assert [n for pt in parts for n in pt[0]] == list(names)
        assert b"".join(bytes(pt[1]) for pt in parts) == bytes(seq)
        assert np.array_equal(np.concatenate([np.diff(pt[2]) for pt in parts]), np.diff(offsets))
        subs = [sharding.contig_subset(offsets, r, world) for r in range(world)]
        assert subs[0][0] == 0 and subs[-1][1] == len(names)
        assert all(subs[r][1] == subs[r + 1][0] for r in range(world - 1))
    with pytest.raises(ValueError):
        sequence.record_aligned_range(p, 2, 2)
    with pytest.raises(ValueError):
        sequence.read_fasta_packed(os.path.join(os.path.dirname(__file__), "golden", "fasta_fixture.fna.gz"), True, (0, 10))


def _gloo_contig_worker(rank, world, port, path, q):
    from tests.gloo_comm import GlooComm
    comm = GlooComm(rank, world, port)
    # two pieces per rank, handed over out of order: rank 0 must put them back into file order
    parts = []
    for k in (1, 0):
        names, seq, offsets = sequence.read_fasta_packed(path, True, sequence.record_aligned_range(path, rank, world, k, 2))
        scores, ids = _fake_contig_scores(seq, offsets)
        parts.append((rank * 64 + k, names, scores, ids))
    out = sharding.gather_contig_parts(comm, parts)
    assert out[3] > 0
    if rank == 0:
        q.put(out)
    else:
        assert out[0] is None and out[1] is None and out[2] is None
    comm.close()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_contig_sharded_front_end_equals_single_process(tmp_path, world):
    """N>1 path of main()'s device front end on CPU (gloo): ranks pack their own byte range, "classify"
    their contigs, rank 0 gathers names / per-contig scores / window ids — equal to one process."""
    mp = pytest.importorskip("torch.multiprocessing")
    import socket
    p = tmp_path / "meta.fna"
    _write_sharding_fasta(p)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_contig_worker, args=(r, world, port, str(p), q)) for r in range(world)]
    for pr in procs:
        pr.start()
    names, preds, ids, total = q.get(timeout=180)
    for pr in procs:
        pr.join(timeout=180)
        assert pr.exitcode == 0
    n1, s1, o1 = sequence.read_fasta_packed(p)
    want_scores, want_ids = _fake_contig_scores(s1, o1)
    assert list(names) == list(n1) and np.array_equal(preds, want_scores)
    assert np.array_equal(ids, want_ids) and total == len(want_ids)


def test_packed_reader_and_candidate_spans_match_reference_rules(golden_dir, tmp_path):
    """read_fasta_packed + candidate_spans (the host half of the contig front end) reproduce what the
    reference's read_fasta / seq_windows yield (goldens generated from the reference itself)."""
    g = json.load(open(os.path.join(golden_dir, "fasta_golden.json")))
    path = os.path.join(golden_dir, "fasta_fixture.fna.gz")
    names, seq, offsets = sequence.read_fasta_packed(path)
    assert list(names) == g["all"]["contig_names"]
    ref = list(sequence.read_fasta(path, strip_n=True))
    assert [len(s) for _, s in ref] == list(np.diff(offsets))
    assert bytes(seq) == "".join(s for _, s in ref).encode()
    for single, key in ((False, "all"), (True, "single")):
        starts, lens, ids, window_n = sequence.candidate_spans(offsets, single)
        # apply the N rule on the host here (the product does it on the device) and compare
        keep = [(wn == 0) or (int(np.count_nonzero(seq[a:a + ln] == 78)) <= 4000)
                for a, ln, wn in zip(starts, lens, window_n)]
        assert list(ids[keep]) == g[key]["contig_ids"]
        wins = sequence.encode_fasta(path, single)[2]
        got = [bytes(seq[a:a + ln]).upper().ljust(6000, b"N") for a, ln, k in zip(starts, lens, keep) if k]
        assert got == [bytes(w) for w in wins]
    # CRLF line ends, text before the first header, '>' inside a line (not a header)
    p = tmp_path / "crlf.fna"
    p.write_bytes(b"junk\r\n>a x\r\nACGT\r\nAC>GT\r\n>b\r\nNNACGTNN\r\n")
    names, seq, offsets = sequence.read_fasta_packed(p)
    assert list(names) == ["a", "b"] and bytes(seq) == b"ACGTAC>GTACGT" and list(offsets) == [0, 9, 13]
    assert [(h, s) for h, s in sequence.read_fasta(p, strip_n=True)] == [("a x", "ACGTAC>GT"), ("b", "ACGT")]
    for L in (1, 2499, 2500, 6000, 8499, 8500, 14499, 14500, 20500):
        st, ln, _, _ = sequence.candidate_spans(np.array([0, L]))
        assert list(ln) == [b - a for a, b in sequence.window_spans(L)]


def test_native_fasta_packer_equals_the_python_rules_and_the_reference(tmp_path):
    """gnn_fasta_scan / gnn_fasta_pack (host code in the library, in place) against the pure-Python
    statement of the rules and, when the checkout is present, against the REFERENCE's own read_fasta
    and check_fasta executed in place — on adversarial layouts and on random FASTA-like text."""
    from oracle import reference_harness
    ref_seq = reference_harness.load_reference_sequence() if reference_harness.available() else None
    cases = [
        b"", b"\n", b">", b">\n", b">a", b">a\n", b">a\nACGT", b">a\nACGT\n", b"ACGT\n>a\nAC\n\nGT\n",
        b">a\n\n\n>b\nNNNN\n>c\nnNACGTNn\n>d\nN\nA\nN\n", b">a desc here\n AC GT \n>b\tq\nAC>GT\n>>c\n>\nAC\n",
        b"\n>a\nACGT\n>a\nACGT\n", b">a\nNNNN\n>a\nNN\n", b">x\n" + b"N" * 5000 + b"\n" + b"ACGT" * 10 + b"\n" + b"n" * 70 + b"\n",
        b">a\r\nAC\r\nGT\r\n>b\rACGT\r", b">only header no newline", b"no header at all\nACGT\n",
        b">a\nACGT\n>b\n>c\n\n>d\nTTTT",
    ]
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"ACGTNnacgt>>\n\n\n xyz", dtype=np.uint8)
    for k in range(60):
        body = alphabet[rng.integers(0, len(alphabet), int(rng.integers(1, 400)))].tobytes()
        cases.append((b">r%d\n" % k if k % 3 else b"") + body)
    for i, text in enumerate(cases):
        p = tmp_path / f"case{i}.fna"
        p.write_bytes(text)
        def attempt(f, *a):
            try:
                return f(*a)
            except IndexError:                       # header without an accession: header.split()[0] fails,
                return "IndexError"                  # in the reference too (sequence.py:24-25)

        for strip in (True, False):
            want = attempt(sequence.read_fasta_packed_py, p, strip)
            got = attempt(sequence.read_fasta_packed, p, strip)
            if isinstance(want, str) or isinstance(got, str):
                assert want == got, (i, text)
                continue
            (got_n, got_s, got_o), (want_n, want_s, want_o) = got, want
            assert list(got_n) == list(want_n), (i, text)
            assert bytes(got_s) == bytes(want_s) and list(got_o) == list(want_o), (i, text)
            rec = list(sequence.read_fasta(p, strip_n=strip))
            assert [len(s) for _, s in rec] == list(np.diff(got_o)), (i, text)
            if ref_seq is not None:
                ref = [(r.accession, r.seq) for r in ref_seq.read_fasta(p, strip_n=strip)]
                assert [a for a, _ in ref] == list(got_n), (i, text)
                assert "".join(s for _, s in ref).encode() == bytes(got_s), (i, text)
        ok = attempt(sequence.check_fasta, p)
        assert ok == attempt(sequence.check_fasta_py, p), (i, text)
        if ref_seq is not None and not isinstance(ok, str):
            assert ok == ref_seq.check_fasta(p), (i, text)


def test_keras_h5_weight_ingestion_roundtrip(synth_weights, tmp_path):
    """a Keras-legacy-shaped HDF5 file (layer groups, ':0' suffixes, nested encoder
    model, an optimizer group that must be ignored) is read back into the repo schema by name + shape."""
    h5 = pytest.importorskip("genomad_amd.h5weights")
    try:
        h5._h5()
    except RuntimeError as exc:
        pytest.skip(str(exc))
    w = synth_weights
    enc = "model_weights/functional_1"          # the frozen encoder is a nested model (model.py:34-38)
    layout = {}

    def put(group, **tensors):
        for k, v in tensors.items():
            layout[f"{group}/{k}:0"] = v
    put(f"{enc}/conv1d/conv1d", kernel=w["conv1_kernel"], bias=w["conv1_bias"])
    put(f"{enc}/conv1d_1/conv1d_1", kernel=w["conv2_kernel"], bias=w["conv2_bias"])
    put(f"{enc}/conv1d_2/conv1d_2", kernel=w["conv3_kernel"], bias=w["conv3_bias"])
    for g, hname in ((f"{enc}/igloo1d_kernel/igloo1d_kernel", "iglooA"), (f"{enc}/igloo1d_kernel_1/igloo1d_kernel_1", "iglooB")):
        put(g, random_patches=w[f"{hname}_patches"], w_mult=w[f"{hname}_w_mult"], w_summer=w[f"{hname}_w_summer"],
            w_bias=w[f"{hname}_w_bias"], w_qk=w[f"{hname}_w_qk"], w_v=w[f"{hname}_w_v"])
    put(f"{enc}/dense/dense", kernel=w["enc_dense_kernel"], bias=w["enc_dense_bias"])
    put(f"{enc}/batch_normalization/batch_normalization", gamma=w["enc_bn_gamma"], beta=w["enc_bn_beta"],
        moving_mean=w["enc_bn_mean"], moving_variance=w["enc_bn_var"])
    put("model_weights/dense_1/dense_1", kernel=w["head_dense_kernel"], bias=w["head_dense_bias"])
    put("model_weights/batch_normalization_1/batch_normalization_1", gamma=w["head_bn_gamma"], beta=w["head_bn_beta"],
        moving_mean=w["head_bn_mean"], moving_variance=w["head_bn_var"])
    put("model_weights/dense_2/dense_2", kernel=w["out_dense_kernel"], bias=w["out_dense_bias"])
    layout["optimizer_weights/adam/iteration:0"] = np.array([7], dtype=np.int32)
    layout["optimizer_weights/adam/dense_2_kernel_momentum:0"] = np.zeros((512, 3), np.float32)
    path = tmp_path / "nn_classifier.h5"
    h5.write_datasets(path, layout)
    got = h5.load_h5(path)
    assert set(got) == set(w)
    for k in w:
        assert np.array_equal(got[k], w[k]), k
    h5.convert(path, tmp_path / "w.npz")
    from genomad_amd import weights as W
    back = W.load_npz(tmp_path / "w.npz")
    assert all(np.array_equal(back[k], w[k]) for k in w)
    # a file with a missing tensor is rejected with a message, not silently accepted
    bad = dict(layout)
    del bad[f"{enc}/conv1d_2/conv1d_2/kernel:0"]
    h5.write_datasets(tmp_path / "bad.h5", bad)
    with pytest.raises(ValueError, match="conv"):
        h5.load_h5(tmp_path / "bad.h5")


def _keras_legacy_layers(w, suffix=":0", swap_convs=False):
    """The reference classifier (model.py:34-45) as Keras' legacy HDF5 saver lays it out: the encoder is a
    nested Model -> ONE group holding all its weights, trainable ones first in layer order, then the
    non-trainable ones (random_patches of both IGLOO kernels, igloo.py:129-135; BatchNormalization moving
    statistics); Dense / BatchNormalization of the head are layers of the outer model; layers without weights
    (input, activations, dropout) are listed with empty groups."""
    def igloo(layer, h):
        return [(f"{layer}/{k}{suffix}", w[f"{h}_{k}"]) for k in ("w_mult", "w_summer", "w_bias", "w_qk", "w_v")]
    c2, c3 = ("conv1d_2", "conv1d_1") if swap_convs else ("conv1d_1", "conv1d_2")
    enc = ([(f"conv1d/kernel{suffix}", w["conv1_kernel"]), (f"conv1d/bias{suffix}", w["conv1_bias"])]
           + igloo("igloo1d_kernel", "iglooA")
           + [(f"{c2}/kernel{suffix}", w["conv2_kernel"]), (f"{c2}/bias{suffix}", w["conv2_bias"]),
              (f"{c3}/kernel{suffix}", w["conv3_kernel"]), (f"{c3}/bias{suffix}", w["conv3_bias"])]
           + igloo("igloo1d_kernel_1", "iglooB")
           + [(f"dense/kernel{suffix}", w["enc_dense_kernel"]), (f"dense/bias{suffix}", w["enc_dense_bias"]),
              (f"batch_normalization/gamma{suffix}", w["enc_bn_gamma"]), (f"batch_normalization/beta{suffix}", w["enc_bn_beta"]),
              (f"igloo1d_kernel/random_patches{suffix}", w["iglooA_patches"]),
              (f"igloo1d_kernel_1/random_patches{suffix}", w["iglooB_patches"]),
              (f"batch_normalization/moving_mean{suffix}", w["enc_bn_mean"]),
              (f"batch_normalization/moving_variance{suffix}", w["enc_bn_var"])])
    return [("input_2", []), ("model", enc),
            ("dense_1", [(f"dense_1/kernel{suffix}", w["head_dense_kernel"]), (f"dense_1/bias{suffix}", w["head_dense_bias"])]),
            ("batch_normalization_1", [(f"batch_normalization_1/gamma{suffix}", w["head_bn_gamma"]),
                                       (f"batch_normalization_1/beta{suffix}", w["head_bn_beta"]),
                                       (f"batch_normalization_1/moving_mean{suffix}", w["head_bn_mean"]),
                                       (f"batch_normalization_1/moving_variance{suffix}", w["head_bn_var"])]),
            ("activation_1", []), ("dropout_2", []),
            ("dense_2", [(f"dense_2/kernel{suffix}", w["out_dense_kernel"]), (f"dense_2/bias{suffix}", w["out_dense_bias"])])]


def test_keras_legacy_saver_layout_and_attribute_cross_check(synth_weights, tmp_path):
    """A file laid out like Keras' legacy saver writes it (root attribute layer_names, per-layer weight_names,
    nested encoder group; weight names with ':0' as Keras 2 writes them and without as Keras 3 does) is read
    into the schema, and the order Keras' own loader would follow (layer_names / weight_names) is cross-checked
    against the name + shape matching: a file whose attribute order contradicts the layer-name order, a dataset
    Keras would not load, and a listed-but-absent weight are all rejected loudly."""
    h5 = pytest.importorskip("genomad_amd.h5weights")
    try:
        h5._h5()
    except RuntimeError as exc:
        pytest.skip(str(exc))
    w = synth_weights
    for suffix in (":0", ""):
        path = tmp_path / f"nn_classifier{len(suffix)}.h5"
        h5.write_keras_legacy(path, _keras_legacy_layers(w, suffix))
        order = h5.read_keras_order(path)
        assert order[0] == f"model/conv1d/kernel{suffix}" and order[-1] == f"dense_2/dense_2/bias{suffix}" and len(order) == 32
        got = h5.load_h5(path)
        assert set(got) == set(w) and all(np.array_equal(got[k], w[k]) for k in w)
    # conv2's tensors stored under the name conv1d_2 and conv3's under conv1d_1, while the attribute order still
    # lists conv2's first: name order and Keras' order disagree -> refuse
    bad = tmp_path / "swapped.h5"
    h5.write_keras_legacy(bad, _keras_legacy_layers(w, ":0", swap_convs=True))
    with pytest.raises(ValueError, match="disagrees"):
        h5.load_h5(bad)
    # a look-alike dataset that is not in weight_names (Keras would never load it)
    layers = _keras_legacy_layers(w)
    stray = tmp_path / "stray.h5"
    h5.write_datasets(stray, {**{f"{ln}/{wn}": a for ln, ws in layers for wn, a in ws},
                              "model/conv1d_9/kernel:0": w["conv3_kernel"]},
                      _attrs=[("", "layer_names", [ln for ln, _ in layers])] +
                             [(ln, "weight_names", [wn for wn, _ in ws]) for ln, ws in layers],
                      _groups=[ln for ln, _ in layers])
    with pytest.raises(ValueError, match="weight_names"):
        h5.load_h5(stray)
    # weight_names lists a weight whose dataset is missing
    layers = _keras_legacy_layers(w)
    short = tmp_path / "short.h5"
    h5.write_datasets(short, {f"{ln}/{wn}": a for ln, ws in layers for wn, a in ws if wn != "dense_2/bias:0"},
                      _attrs=[("", "layer_names", [ln for ln, _ in layers])] +
                             [(ln, "weight_names", [wn for wn, _ in ws]) for ln, ws in layers],
                      _groups=[ln for ln, _ in layers])
    with pytest.raises(ValueError):
        h5.load_h5(short)
    # files without the attributes (not written by Keras) still load by name + shape alone
    plain = tmp_path / "plain.h5"
    h5.write_datasets(plain, {f"{ln}/{wn}": a for ln, ws in layers for wn, a in ws})
    assert h5.read_keras_order(plain) is None
    assert all(np.array_equal(h5.load_h5(plain)[k], w[k]) for k in w)


def test_h5weights_cli_dry_run_prints_the_assignment_and_names_the_failing_rule(synth_weights, tmp_path, capsys):
    """`python -m genomad_amd.h5weights convert --dry-run nn_classifier.h5`: the one command the first person with the real
    blob runs.  It prints dataset -> schema tensor for all 36 tensors (and whether Keras' own load order was available for
    the cross-check), writes nothing; on a file that does not match it exits 2 with the rule that failed."""
    h5 = pytest.importorskip("genomad_amd.h5weights")
    try:
        h5._h5()
    except RuntimeError as exc:
        pytest.skip(str(exc))
    w = synth_weights
    good = tmp_path / "nn_classifier.h5"
    h5.write_keras_legacy(good, _keras_legacy_layers(w))
    assert h5._main(["convert", "--dry-run", str(good)]) == 0
    out = capsys.readouterr().out
    assert "cross-checked against the order Keras loads in" in out and "dry run: nothing written" in out
    assert "conv3_kernel" in out and "<- model/conv1d_2/kernel:0" in out and "<- model/igloo1d_kernel_1/w_qk:0" in out
    assert sum(1 for line in out.splitlines() if " <- " in line and not line.startswith("schema tensor")) == len(w)
    assert not list(tmp_path.glob("*.npz"))
    assert h5._main(["convert", str(good), str(tmp_path / "w.npz")]) == 0 and (tmp_path / "w.npz").exists()
    capsys.readouterr()
    bad = tmp_path / "swapped.h5"
    h5.write_keras_legacy(bad, _keras_legacy_layers(w, ":0", swap_convs=True))
    assert h5._main(["convert", "--dry-run", str(bad)]) == 2
    err = capsys.readouterr().err
    assert "does not look like the nn_classifier.h5" in err and "disagrees with the order" in err


def test_gen_patches_follows_the_reference_initializer():
    """synthetic.gen_patches restates gen_filters_igloo(4, 2100, 5997, return_sequences=False,
    build_backbone=False) (igloo.py:220-302, the initializer of `random_patches`, :106-115).  The reference
    function is pure numpy, so it is RUN IN PLACE here: same shape and dtype kind, every patch = 4 distinct
    positions in [0, 5997) sorted ascending, and — both being uniform draws — matching position statistics."""
    from oracle import reference_harness as rh
    if not rh.available():
        pytest.skip("needs the reference checkout")
    from genomad_amd import synthetic
    nn = rh.load_reference_network()
    import importlib
    igloo = importlib.import_module("genomad.neural_network.igloo")
    np.random.seed(5)
    ref = np.asarray(igloo.gen_filters_igloo(4, 2100, 5997, return_sequences=False, build_backbone=False))
    mine = synthetic.gen_patches(np.random.default_rng(5))
    assert ref.shape == mine.shape == (2100, 4, 1)
    for p in (ref.astype(np.int64), mine.astype(np.int64)):
        flat = p[:, :, 0]
        assert flat.min() >= 0 and flat.max() < 5997
        assert (np.diff(flat, axis=1) > 0).all()                 # 4 DISTINCT positions, ascending
    r, m = ref[:, :, 0].astype(np.float64), mine[:, :, 0].astype(np.float64)
    # order statistics of 4 uniform draws without replacement from [0, 5997): E = 5997 * k / 5
    for k in range(4):
        expect = 5997 * (k + 1) / 5
        assert abs(r[:, k].mean() - expect) < 120 and abs(m[:, k].mean() - expect) < 120
    assert abs(r.std() - m.std()) < 60
    assert nn is not None


def test_tfrecord_wire_format(tmp_path):
    """TFRecord framing + tf.train.Example encoding without TensorFlow."""
    from genomad_amd import _lib, synthetic, tfrecord
    lib = _lib.load()
    assert lib.gnn_crc32c(b"123456789", 9) == 0xE3069283            # CRC-32C check value
    assert lib.gnn_crc32c(b"", 0) == 0
    # the serialisation of Example{features{feature{"sequence": int64_list[1,2,3]}}}
    assert tfrecord.encode_example(np.array([1, 2, 3])) == \
        b"\n\x15\n\x13\n\x08sequence\x12\x07\x1a\x05\n\x03\x01\x02\x03"
    assert tfrecord.encode_example(np.array([0, 127, 128, 256])).endswith(b"\n\x06\x00\x7f\x80\x01\x80\x02")
    bases = synthetic.synth_windows(0, 12)
    toks = sequence_oracle.tokenize_closed_form(bases)
    p = tmp_path / "12.tfrec"
    tfrecord.write_file(p, toks)
    raw = p.read_bytes()
    n0 = int.from_bytes(raw[:8], "little")
    assert len(tfrecord.encode_example(toks[0])) == n0 and len(raw) > 12 * (5997 + 16)
    assert np.array_equal(tfrecord.read_file(p), toks)
    corrupt = bytearray(raw)
    corrupt[40] ^= 1
    (tmp_path / "bad.tfrec").write_bytes(bytes(corrupt))
    with pytest.raises(ValueError, match="corrupt"):
        tfrecord.read_file(tmp_path / "bad.tfrec")
    # file naming of generate_data: 10 000 per file, named by the running count
    monkey = tfrecord.RECORDS_PER_FILE
    tfrecord.RECORDS_PER_FILE = 5
    try:
        files = tfrecord.write_dir(tmp_path, toks)
        assert [f.name for f in files] == ["5.tfrec", "10.tfrec", "12.tfrec"]
        (tmp_path / "bad.tfrec").unlink()
        assert np.array_equal(tfrecord.read_dir(tmp_path), toks)     # 5 + 5 + 2 records, numeric file order
    finally:
        tfrecord.RECORDS_PER_FILE = monkey
    # tokens -> bases -> tokens is the identity (windows 5 and 9 contain N padding / an N run)
    back = tfrecord.tokens_to_bases(toks)
    assert back.shape == (12, 6000)
    assert np.array_equal(sequence_oracle.tokenize_closed_form(back), toks)


def test_resume_from_a_reference_encoded_directory(tmp_path):
    """An `_encoded_sequences` directory as the reference leaves it (``*.tfrec`` + seq_window_id.npz,
    no .win.npy) is picked up on resume: tokens are turned back into equivalent windows."""
    from genomad_amd import tfrecord
    rng = np.random.default_rng(2)
    fa = tmp_path / "r.fna"
    _write_fasta(fa, [("k1", "".join(rng.choice(list("ACGTN"), 9000, p=[.24, .24, .24, .24, .04]))), ("k2", "ACGT" * 700)])
    out = tmp_path / "out"
    nnc.main(fa, out, False, 128, False, 1, False, False, _backend=FakeBackend())
    d = out / "r_nn_classification"
    first = np.load(d / "r_nn_classification.npz")["predictions"]
    enc = d / "r_encoded_sequences"
    (win,) = list(enc.glob("*.win.npy"))
    windows = np.load(win)
    tfrecord.write_dir(enc, sequence_oracle.tokenize_closed_form(windows))     # what the reference would have written
    win.unlink()
    (d / "r_nn_classification.npz").unlink()

    class Check(FakeBackend):
        def score(self, w):
            # equivalent windows: identical tokens
            assert np.array_equal(sequence_oracle.tokenize_closed_form(w), sequence_oracle.tokenize_closed_form(windows))
            return FakeBackend.score(self, windows)
    nnc.main(fa, out, False, 128, False, 1, False, False, _backend=Check())
    assert np.array_equal(np.load(d / "r_nn_classification.npz")["predictions"], first)


# ------------------------------------------------------------------ the drop-in seam (cli.py:772-774, :1367-1376)
def test_install_rebinds_the_reference_entry_point_and_undoes_cuda_visible_devices(tmp_path, monkeypatch):
    """The reference CLI calls ``genomad.nn_classification.main(input, output, single_window, batch_size,
    restart, threads, verbose, cleanup)`` positionally (cli.py:772-774; end-to-end goes through the same
    command, :1367-1376).  Importing the reference module exports CUDA_VISIBLE_DEVICES=-1
    (modules/nn_classification.py:8), which HIP honours: install() and main() must remove it before any HIP
    call.  Runs the REAL reference package namespace in place (numba stubbed, genomad/__init__.py bypassed)."""
    from oracle import reference_harness as rh
    if not rh.available():
        pytest.skip("needs the reference checkout")
    import importlib
    rh.load_reference_sequence()
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    ref_mod = importlib.import_module("genomad.modules.nn_classification")
    importlib.reload(ref_mod)                                   # (re)runs its module-level os.environ writes
    assert os.environ.get("CUDA_VISIBLE_DEVICES") == "-1"
    import genomad
    monkeypatch.setattr(genomad, "nn_classification", ref_mod, raising=False)   # what genomad/__init__.py:5-14 binds
    monkeypatch.setitem(sys.modules, "genomad.modules.nn_classification", ref_mod)
    try:
        this = nnc.install()
        assert this is nnc and genomad.nn_classification is nnc
        assert sys.modules["genomad.modules.nn_classification"] is nnc
        assert "CUDA_VISIBLE_DEVICES" not in os.environ
        os.environ["CUDA_VISIBLE_DEVICES"] = "-1"              # e.g. another module re-imported the reference one
        monkeypatch.setenv("GENOMAD_AMD_FRONT_END", "host")
        monkeypatch.delenv("WORLD_SIZE", raising=False)
        monkeypatch.setattr(nnc, "GpuBackend", lambda batch_size, console=None: FakeBackend())

        def cli_nn_classification(input, output, single_window, batch_size, restart, threads, verbose, cleanup):
            genomad.nn_classification.main(input, output, single_window, batch_size, restart, threads, verbose, cleanup)

        from pathlib import Path
        fa = tmp_path / "sample.fna"
        _write_fasta(fa, [("c1 d", "ACGT" * 3000), ("c2", "GGCA" * 700)])
        cli_nn_classification(Path(fa), Path(tmp_path / "out"), False, 128, False, 1, False, False)
        assert "CUDA_VISIBLE_DEVICES" not in os.environ
        z = np.load(tmp_path / "out" / "sample_nn_classification" / "sample_nn_classification.npz")
        assert list(z["contig_names"]) == ["c1", "c2"] and z["predictions"].shape == (2, 3)
        # end-to-end's keyword form resolves to the same positional call (click fills the defaults)
        cli_nn_classification(input=Path(fa), output=Path(tmp_path / "out2"), single_window=False, batch_size=128,
                              restart=False, threads=1, verbose=False, cleanup=True)
        assert (tmp_path / "out2" / "sample_nn_classification" / "sample_nn_classification.tsv").exists()
    finally:
        os.environ.pop("CUDA_VISIBLE_DEVICES", None)
        sys.modules["genomad.modules.nn_classification"] = ref_mod
        import genomad.modules
        genomad.modules.nn_classification = ref_mod


def _gloo_main_worker(rank, world, port, fasta, out_dir, q):
    os.environ["CUDA_VISIBLE_DEVICES"] = "-1"                  # as left behind by the reference module's import
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from tests.gloo_comm import GlooComm
    comm = GlooComm(rank, world, port)
    for attempt in range(2):                                    # second pass: the resume path, decided on rank 0
        nnc.main(fasta, out_dir, False, 128, False, 1, False, False, _backend=FakeBackend(), _comm=comm)
        assert "CUDA_VISIBLE_DEVICES" not in os.environ
    q.put(rank)
    comm.close()


def test_main_under_world_size_2_equals_single_process(tmp_path):
    """main() with two ranks (gloo transport, fake scorer): rank 0 writes the same files as one process,
    no rank hangs on the resume pass although only rank 0 reads the files that decide it, and the
    CUDA_VISIBLE_DEVICES=-1 the reference leaves behind is gone before any device work."""
    mp = pytest.importorskip("torch.multiprocessing")
    import socket
    fa = tmp_path / "meta.fna"
    _write_sharding_fasta(fa)
    nnc.main(fa, tmp_path / "single", False, 128, False, 1, False, False, _backend=FakeBackend())
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_main_worker, args=(r, 2, port, str(fa), str(tmp_path / "multi"), q)) for r in range(2)]
    for p in procs:
        p.start()
    done = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert done == [0, 1]
    a = np.load(tmp_path / "single" / "meta_nn_classification" / "meta_nn_classification.npz")
    b = np.load(tmp_path / "multi" / "meta_nn_classification" / "meta_nn_classification.npz")
    assert list(a["contig_names"]) == list(b["contig_names"]) and np.array_equal(a["predictions"], b["predictions"])
    assert ((tmp_path / "single" / "meta_nn_classification" / "meta_nn_classification.tsv").read_text()
            == (tmp_path / "multi" / "meta_nn_classification" / "meta_nn_classification.tsv").read_text())
    assert not list((tmp_path / "multi" / "meta_nn_classification").glob("*.tmp*"))


def test_check_fasta_in_chunks_and_non_ascii_headers(tmp_path):
    p = tmp_path / "u.fna"
    p.write_bytes(">caf\xc3\xa9 one\nACGT\n>b\xff raw\nGG\nTT\n>c\nAC\n".replace("\\", "\\").encode("latin-1"))
    names, seq, offsets = sequence.read_fasta_packed(p)
    assert names[0] == "caf\u00e9" and len(names) == 3 and bytes(seq) == b"ACGTGGTTAC"
    assert sequence.check_fasta(p) and sequence.check_fasta(p, chunk_bytes=8)
    dup = tmp_path / "d.fna"
    dup.write_text(">a\nAC\n" * 1 + ">b\nGG\n" * 50 + ">a\nTT\n")
    assert not sequence.check_fasta(dup) and not sequence.check_fasta(dup, chunk_bytes=16)
    big = tmp_path / "meta.fna"
    _write_sharding_fasta(big)
    whole = sequence.read_fasta_packed(big)
    parts = [sequence.pack_text(c) for c in sequence.iter_text_chunks(big, 20000)]
    assert len(parts) > 3 and [n for pt in parts for n in pt[0]] == list(whole[0])
    assert b"".join(bytes(pt[1]) for pt in parts) == bytes(whole[1])


# ------------------------------------------------------------------ RCCL bootstrap without GPUs
class _FakeCommLib:
    """Stands in for libgenomad_nn_hip.so's gnn_comm_* (tests only): records what the bootstrap hands to
    ncclCommInitRank and implements the collectives over a shared directory."""

    def __init__(self, box):
        self.box = box

    def gnn_comm_unique_id(self, uid):
        import ctypes as C
        C.memmove(uid, bytes((i * 7 + 3) % 256 for i in range(128)), 128)
        return 0

    def gnn_comm_init(self, ctx, world, rank, uid):
        # like ncclCommInitRank, returns only once every rank has joined
        import time
        from pathlib import Path
        self.box["init"] = (world, rank, bytes(uid))
        side = Path(self.box["side"])
        (side / f"joined_{rank}").write_text("x")
        deadline = time.time() + 30
        while len(list(side.glob("joined_*"))) < world:
            assert time.time() < deadline
            time.sleep(0.01)
        return 0

    def gnn_comm_barrier(self, ctx):
        self.box["barriers"] = self.box.get("barriers", 0) + 1
        return 0

    def gnn_comm_destroy(self, ctx):
        return 0


def _rccl_bootstrap_worker(rank, world, port, rdzv_dir, q):
    import time
    os.environ.update(MASTER_PORT=str(port), GENOMAD_AMD_RDZV_DIR=rdzv_dir, TORCHELASTIC_RUN_ID="t1")
    from genomad_amd import rccl
    box = {"side": os.path.join(os.path.dirname(rdzv_dir), "side")}
    eng = type("E", (), {"lib": _FakeCommLib(box), "ctx": object()})()
    if rank == 0:
        time.sleep(0.3)                       # the other ranks are already polling for the id file
    comm = rccl.RcclComm(eng, rank, world, timeout=30)
    q.put((rank, box["init"], box["barriers"], sorted(os.listdir(rdzv_dir)) if rank == 0 else None))
    comm.close()


def test_rccl_unique_id_bootstrap_with_three_ranks(tmp_path):
    """The part of the multi-GPU path that cannot run on a one-GPU box: rank 0 publishes ncclGetUniqueId's 128
    bytes in a file keyed by launcher port / run id / parent pid, the other ranks wait for it, every rank calls
    ncclCommInitRank with the same id and its own rank, and rank 0 removes the file after the first barrier."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    d = tmp_path / "rdzv"
    d.mkdir()
    (tmp_path / "side").mkdir()
    # leftovers of a crashed run under the very name this launch will use: a bare 128-byte id (the round-2 format), then a
    # well-formed record of ANOTHER launch (wrong tag) - the polling ranks must accept neither, rank 0 replaces the file
    import struct
    import time as _time
    from genomad_amd import rccl
    env = dict(MASTER_PORT="29999", GENOMAD_AMD_RDZV_DIR=str(d), TORCHELASTIC_RUN_ID="t1", GENOMAD_AMD_RDZV_PARENT=str(os.getpid()))
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        stale = rccl._id_file(0)
        tag = rccl._run_tag(0)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    assert stale.parent == d
    stale.write_bytes(b"\x55" * 128)
    assert rccl._read_id(stale, tag, _time.time()) is None
    stale.write_bytes(rccl._MAGIC + bytes(16) + b"\x55" * 128 + struct.pack("<d", _time.time()))
    assert rccl._read_id(stale, tag, _time.time()) is None
    stale.write_bytes(rccl._MAGIC + tag + b"\x55" * 128 + struct.pack("<d", _time.time() - 86400))     # right tag, a day old
    assert rccl._read_id(stale, tag, _time.time()) is None
    stale.write_bytes(rccl._MAGIC + bytes(16) + b"\x55" * 128 + struct.pack("<d", _time.time()))
    procs = [ctx.Process(target=_rccl_bootstrap_worker, args=(r, 3, 29999, str(d), q)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_id = bytes((i * 7 + 3) % 256 for i in range(128))
    assert [g[1] for g in got] == [(3, r, want_id) for r in range(3)]
    assert all(g[2] >= 1 for g in got)
    assert got[0][3] == []                     # the id file is gone once everybody has joined


# ------------------------------------------------------------------ bench.py as its own launcher
from pathlib import Path  # noqa: E402

_SPAWN_WORKER = '''
import json, os, sys, time
sys.path.insert(0, {root!r})
from tests.test_host import _FakeCommLib
from genomad_amd import rccl
rank, world, local = rccl.world_from_env()
box = {{"side": os.environ["SIDE"]}}
eng = type("E", (), {{"lib": _FakeCommLib(box), "ctx": object()}})()
comm = rccl.RcclComm(eng, rank, world, timeout=30)
if os.environ.get("FAIL_RANK") == str(rank):
    sys.exit(7)
if os.environ.get("FAIL_RANK") is not None:
    time.sleep(120)          # the launcher has to stop this rank when the failing one exits
print(json.dumps({{"rank": rank, "world": world, "local": local, "id": box["init"][2].hex()[:8],
                  "rdzv": os.environ["GENOMAD_AMD_RDZV_DIR"]}}), flush=True)
'''


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` from a plain process (no launcher environment) has to become N ranks itself: RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* exported, a private rendezvous directory for the RCCL unique id, rank 0's stdout =
    the output, the first failing rank's exit code returned and the others stopped.  The ranks here bootstrap RcclComm over
    the fake comm library above (no GPU)."""
    import subprocess
    import time
    root = str(Path(__file__).resolve().parents[1])
    worker = tmp_path / "worker.py"
    worker.write_text(_SPAWN_WORKER.format(root=root))
    (tmp_path / "side").mkdir()
    launch = (f"import sys; sys.path.insert(0, {root!r}); import bench; "
              f"sys.exit(bench.spawn_ranks(3, [sys.executable, {str(worker)!r}]))")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "FAIL_RANK")}
    env["SIDE"] = str(tmp_path / "side")
    r = subprocess.run([sys.executable, "-c", launch], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1 and lines[0]["rank"] == 0 and lines[0]["world"] == 3      # only rank 0 owns stdout
    others = sorted(json.loads(x)["rank"] for x in r.stderr.splitlines() if x.startswith("{"))
    assert others == [1, 2]
    assert not os.path.exists(lines[0]["rdzv"])                                      # the private directory is removed
    # a failing rank: its exit code comes back, and quickly (the sleeping ranks are terminated, not waited for)
    for f in (tmp_path / "side").iterdir():
        f.unlink()
    env["FAIL_RANK"] = "1"
    t = time.time()
    r = subprocess.run([sys.executable, "-c", launch], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 7 and time.time() - t < 60, (r.returncode, r.stderr)


def test_bench_cli_self_launch_reaches_the_ranks(tmp_path):
    """The driver's form, `python bench.py --gpus 2 ...`, in a process without launcher variables: both ranks start and fail
    where a box without GPUs must fail (creating the engine), i.e. after the spawn - not with the round-2 refusal."""
    import subprocess
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""            # no device, whatever the box has: the ranks stop at gnn_create
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-sample", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs a launch with" not in r.stderr
    assert "rank" in r.stderr and "exited with" in r.stderr          # the launcher's report of the first failing rank


def test_unknown_precision_name_is_refused_up_front(tmp_path, monkeypatch):
    """GENOMAD_AMD_PRECISION is the one knob a user of the drop-in sets by hand: a typo must stop main() before it creates the
    output directory's contents, with the valid names in the message."""
    monkeypatch.setenv("GENOMAD_AMD_PRECISION", "f16x4")
    with pytest.raises(ValueError, match="f16x3"):
        nnc.configured_precision()
    fa = tmp_path / "a.fna"
    fa.write_text(">c1\n" + "ACGT" * 800 + "\n")
    with pytest.raises(ValueError, match="GENOMAD_AMD_PRECISION"):
        nnc.main(fa, tmp_path / "out", False, 128, True, 1, False, False)
    assert not (tmp_path / "out" / "a_nn_classification").exists()
    monkeypatch.setenv("GENOMAD_AMD_PRECISION", "bf16x3")
    assert nnc.configured_precision() == "bf16x3"
    monkeypatch.delenv("GENOMAD_AMD_PRECISION")
    assert nnc.configured_precision() == nnc.DEFAULT_PRECISION == "f16x3tc"


def test_bench_n8_line_is_complete_over_the_fake_engine(tmp_path):
    """The driver's own command, `python bench.py --gpus 8 --steps 20 --warmup 5`, end to end on the CPU over tests/fake_engine.py
    (GENOMAD_AMD_BENCH_FAKE_ENGINE=1): bench.py spawns its 8 ranks, shards the job contiguously, gathers the scores on rank 0
    and prints ONE line that carries what an 8-GPU measurement has to be read with - `roofline`, `cpu_baseline` (rank 0, before
    the communicator exists), `rccl_ranks` 8, one `per_rank_*` entry per rank, the gather's own duration and the communicator's
    start-up time - and that passes its own parity checks (the fake serves the committed reference-graph scores)."""
    import subprocess
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["GENOMAD_AMD_BENCH_FAKE_ENGINE"] = "1"
    env["OPENBLAS_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5",
                        "--windows-per-step", "1024", "--cpu-sample", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    o = lines[0]
    assert o["fake_engine"] is True and o["data"].startswith("FAKE ENGINE")
    assert o["n_gpus"] == 8 and o["rccl_ranks"] == 8 and o["steps"] == 20 and o["warmup"] == 5 and o["scaling"] == "strong"
    assert len(o["per_rank_windows_per_s"]) == len(o["per_rank_steps_ms"]) == len(o["per_rank_front_ms_total"]) == 8
    assert o["gather_ms"] > 0 and o["gather_ms_isolated"] > 0 and o["comm_init_s"] > 0
    assert o["roofline"]["kernel"] == "gnn::tc::fused_front_tc_kernel<false>" and o["roofline"]["bound"] == "mfma"
    assert o["cpu_baseline"]["kind"] == "port" and o["cpu_baseline"]["value"] > 0 and o["max_abs_dscore_vs_cpu_baseline"] < 1e-4
    assert o["parity"]["ok"] and o["parity"]["windows"] == 20 * 1024
    assert o["steps_verified"]["mismatching_windows_all_ranks"] == 0
    assert o["max_abs_dscore"] == 0.0 and o["dscore_windows"] == 10000       # rank 0 holds ALL gathered windows, in job order
    assert "failed" not in o


def test_bench_reports_a_dying_rank_with_one_json_line(tmp_path):
    """the first 8-GPU run of bench.py happens unattended.  A rank that dies (here: rank 3 of 4 on entering the
    gather; rank 1 of 2 in the timed region under the driver's launcher) must leave ONE JSON line on stdout with "error", the rank,
    the stage, the RCCL ranks seen and the library's last error, a non-zero exit code, and no hanging peers."""
    import socket
    import subprocess
    import time
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "OMP_NUM_THREADS")}
    env.update(GENOMAD_AMD_BENCH_FAKE_ENGINE="1", OPENBLAS_NUM_THREADS="2", GENOMAD_AMD_BENCH_TEST_FAIL="3:gather")
    t = time.time()
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "4", "--steps", "4", "--warmup", "1", "--windows-per-step", "1024",
                        "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and time.time() - t < 120
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout
    o = lines[0]
    assert o["error"] == "RuntimeError" and o["rank"] == 3 and o["world"] == 4 and o["stage"] == "gather" and o["rccl_ranks"] == 4
    assert o["value"] is None and "gnn_last_error" in o and "GENOMAD_AMD_BENCH_TEST_FAIL" in o["detail"]
    assert "rank 0 FAILED in stage" in r.stderr                              # the peers were stopped, and said where they were
    pytest.importorskip("torch")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env["GENOMAD_AMD_BENCH_TEST_FAIL"] = "1:timed"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--windows-per-step", "512", "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=900, cwd=str(root))
    assert r.returncode != 0
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert lines[0]["rank"] == 1 and lines[0]["stage"] == "timed" and lines[0]["error"] == "RuntimeError"


def test_out_of_tolerance_arithmetics_need_an_explicit_opt_in(monkeypatch):
    """f16c6 is documented to leave the 1e-4 tolerance on 10^6 windows (1.2e-4) and a 64-window sentinel cannot see a
    1-in-10^5 tail - main() refuses it unless GENOMAD_AMD_ALLOW_OUT_OF_TOLERANCE=1, and then says so once per run; the two arithmetics
    round 6 removed (bf16, f16c8) are unknown names."""
    monkeypatch.setenv("GENOMAD_AMD_PRECISION", "f16c6")
    monkeypatch.delenv("GENOMAD_AMD_ALLOW_OUT_OF_TOLERANCE", raising=False)
    with pytest.raises(ValueError, match="GENOMAD_AMD_ALLOW_OUT_OF_TOLERANCE"):
        nnc.configured_precision()
    monkeypatch.setenv("GENOMAD_AMD_ALLOW_OUT_OF_TOLERANCE", "1")
    assert nnc.configured_precision() == "f16c6"
    said = []

    class Console:
        def log(self, msg):
            said.append(msg)
    nnc._WARNED.discard(("oot", "f16c6"))
    nnc._warn_out_of_tolerance(Console(), "f16c6")
    nnc._warn_out_of_tolerance(Console(), "f16c6")
    nnc._warn_out_of_tolerance(Console(), "f16x3tc")
    assert len(said) == 1 and "OUTSIDE the 1e-4" in said[0] and "1.2e-04" in said[0]
    for gone in ("bf16", "f16c8"):
        monkeypatch.setenv("GENOMAD_AMD_PRECISION", gone)
        with pytest.raises(ValueError, match="expected one of"):
            nnc.configured_precision()


# ------------------------------------------------------------------ sharded FASTA validation
class _FakeContigEngine:
    """classify_contigs of the device front end with a fixed function of the contig bytes in place of the network."""

    def classify(self, windows, precision=None):          # the parity sentinel's two calls: the same function for every arithmetic
        return FakeBackend().score(windows)

    def classify_contigs(self, seq, offsets, single_window=False, precision=None):
        offsets = np.asarray(offsets, np.int64)
        _, _, ids, _ = sequence.candidate_spans(offsets, single_window)
        sc = np.zeros((len(offsets) - 1, 3), np.float32)
        for c in range(len(offsets) - 1):
            v = float(np.asarray(seq[offsets[c]:offsets[c + 1]], np.int64).sum() % 997) / 997
            sc[c] = (v, (1 - v) / 2, (1 - v) / 2)
        return sc, ids


def _sharded_check_worker(rank, world, port, fasta, out_dir, q, collide):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.pop("GENOMAD_AMD_FRONT_END", None)
    from genomad_amd import sharding
    from tests.gloo_comm import GlooComm
    nnc._engine = lambda: _FakeContigEngine()
    calls = []
    real_check = sequence.check_fasta
    sequence.check_fasta = lambda p, *a, **k: (calls.append(rank), real_check(p, *a, **k))[1]
    if collide:                                      # every accession gets the same digest: the exact check has to decide
        real = sequence.accession_digests_of_text
        sequence.accession_digests_of_text = lambda text: np.full(len(real(text)), rank, dtype="<u8")   # equal inside every share
    comm = GlooComm(rank, world, port)
    code = 0
    try:
        nnc.main(fasta, out_dir, False, 128, False, 1, False, False, _comm=comm)
    except SystemExit as e:
        code = e.code
    q.put((rank, code, calls))
    comm.close()


def _run_sharded_main(tmp_path, fasta, world, collide=False):
    mp = pytest.importorskip("torch.multiprocessing")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    out = tmp_path / f"out_{world}_{int(collide)}_{fasta.stem}"
    procs = [ctx.Process(target=_sharded_check_worker, args=(r, world, port, str(fasta), str(out), q, collide)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    return out, got


def test_fasta_validation_is_sharded_with_the_contigs(tmp_path, monkeypatch):
    """With several ranks the device front end validates the FASTA where it reads it: every rank collects the accessions of ITS
    records (all of them: a record the N strip drops still counts, as in the reference's check_fasta, sequence.py:124-131), one
    gather of 64-bit digests finds duplicates across ranks, and the sequential whole-file check runs only if two digests are
    equal.  Same verdicts, same files and same exit status as one process; nothing is written for an invalid input."""
    rng = np.random.default_rng(5)
    recs = [(f"c{i} note", "".join(rng.choice(list("ACGT"), int(L)))) for i, L in enumerate(rng.integers(3000, 20000, 14))]
    recs.insert(5, ("alln", "N" * 7000))                                       # dropped by the classification pass
    good = tmp_path / "good.fna"
    _write_fasta(good, recs)
    monkeypatch.setattr(nnc, "_engine", lambda: _FakeContigEngine())
    monkeypatch.delenv("GENOMAD_AMD_FRONT_END", raising=False)
    nnc.main(good, tmp_path / "single", False, 128, False, 1, False, False)
    ref = np.load(tmp_path / "single" / "good_nn_classification" / "good_nn_classification.npz")
    assert "alln" not in list(ref["contig_names"]) and len(ref["contig_names"]) == 14
    for world in (2, 3):
        out, got = _run_sharded_main(tmp_path, good, world)
        assert [g[1] for g in got] == [0] * world
        assert all(g[2] == [] for g in got)                                     # the sequential check never ran
        z = np.load(out / "good_nn_classification" / "good_nn_classification.npz")
        assert list(z["contig_names"]) == list(ref["contig_names"]) and np.array_equal(z["predictions"], ref["predictions"])
        assert json.load(open(out / "good_nn_classification" / "good_nn_classification.json"))["input_md5"] == nnc._md5_of(good)
    # equal digests (forced): the exact check decides, on rank 0 only, and the run still succeeds
    out, got = _run_sharded_main(tmp_path, good, 2, collide=True)
    assert [g[1] for g in got] == [0, 0] and got[0][2] == [0] and got[1][2] == []
    # duplicates: across ranks (first and last record), inside one rank (neighbours), and of a record that is all N
    for name, edit in (("across", lambda r: r + [("c0 again", "ACGT" * 900)]),
                       ("inside", lambda r: r[:2] + [("c1", "GGCC" * 900)] + r[2:]),
                       ("dropped", lambda r: r + [("alln x", "ACGT" * 900)])):
        bad = tmp_path / f"{name}.fna"
        _write_fasta(bad, edit(list(recs)))
        assert not sequence.check_fasta(bad)
        out, got = _run_sharded_main(tmp_path, bad, 3)
        assert [g[1] for g in got] == [1, 1, 1], name                          # every rank leaves through sys.exit(1) (:164-170)
        assert not (out / f"{name}_nn_classification").exists()
    empty = tmp_path / "empty.fna"
    empty.write_text("")
    out, got = _run_sharded_main(tmp_path, empty, 2)
    assert [g[1] for g in got] == [1, 1]


def test_deferred_execution_info_is_opt_in_and_writes_the_same_file(tmp_path, monkeypatch):
    """GENOMAD_AMD_DEFER_EXECUTION_INFO=1: main() returns once the scores are on disk without waiting for the input's md5 (the
    reference's contract, utils.py:216-254, and the longest sequential thing left with 8 GPUs); the JSON is written by the
    background thread when the digest is ready, byte for byte what the default path writes (start time aside)."""
    import threading
    import time
    fa = tmp_path / "a.fna"
    _write_fasta(fa, [("c1", "ACGT" * 2000), ("c2", "GGCA" * 900)])
    release = threading.Event()
    real = nnc._md5_of

    def slow(path, size=1 << 22):
        release.wait(30)
        return real(path, size)
    monkeypatch.setattr(nnc, "_md5_of", slow)
    nnc._MD5_FUTURES.clear()
    monkeypatch.setenv("GENOMAD_AMD_DEFER_EXECUTION_INFO", "1")
    t = time.time()
    nnc.main(fa, tmp_path / "o1", False, 128, True, 1, False, False, _backend=FakeBackend())
    info = tmp_path / "o1" / "a_nn_classification" / "a_nn_classification.json"
    assert time.time() - t < 20 and not info.exists()                      # returned although the digest is still pending
    assert (tmp_path / "o1" / "a_nn_classification" / "a_nn_classification.tsv").exists()
    release.set()
    nnc.wait_execution_info()
    got = json.load(open(info))
    monkeypatch.delenv("GENOMAD_AMD_DEFER_EXECUTION_INFO")
    nnc._MD5_FUTURES.clear()
    nnc.main(fa, tmp_path / "o2", False, 128, True, 1, False, False, _backend=FakeBackend())
    want = json.load(open(tmp_path / "o2" / "a_nn_classification" / "a_nn_classification.json"))
    got.pop("start_time"), want.pop("start_time")
    assert got == want and got["input_md5"] == real(fa)


def test_range_fallback_chain_of_the_default_arithmetic():
    """The Toom-Cook form's transformed activations leave the f16 range before the activations themselves do (|x| > ~2 000 vs
    65 504): non-finite scores from f16x3tc are recomputed with the direct f16x3 form first - f32-class accuracy - and only if that
    is non-finite too with bf16x3 (f32 range).  One log line per step of the chain."""
    calls, logs = [], []

    class Eng:
        def __init__(self, finite):
            self.finite = finite

        def classify_contigs(self, seq, offsets, single_window, precision):
            calls.append(precision)
            pr = np.full((2, 3), 1 / 3, np.float32)
            if precision not in self.finite:
                pr[1, 0] = np.nan
            return pr, np.array([0, 1])

    console = type("C", (), {"log": lambda self, m, **k: logs.append(m)})()
    nnc._WARNED.clear()
    pr, _ = nnc.classify_contigs_safely(Eng({"f16x3", "bf16x3"}), None, None, False, "f16x3tc", console)
    assert calls == ["f16x3tc", "f16x3"] and np.isfinite(pr).all() and len(logs) == 1 and "with f16x3." in logs[0]
    calls.clear()
    pr, _ = nnc.classify_contigs_safely(Eng({"bf16x3"}), None, None, False, "f16x3tc", console)
    assert calls == ["f16x3tc", "f16x3", "bf16x3"] and np.isfinite(pr).all() and len(logs) == 2 and "with bf16x3." in logs[1]
    calls.clear()
    pr, _ = nnc.classify_contigs_safely(Eng({"f16x3tc"}), None, None, False, "f16x3tc", console)
    assert calls == ["f16x3tc"] and len(logs) == 2
    calls.clear()
    nnc.classify_contigs_safely(Eng(set()), None, None, False, "f32", console)       # exact f32: nothing to fall back to
    assert calls == ["f32"]


# ------------------------------------------------------------------ runtime parity sentinel
class _SentinelEngine(_FakeContigEngine):
    """classify() = the fake score function, off by `delta` for every arithmetic but f32; non-finite for the arithmetics in `nan`."""

    def __init__(self, delta=0.0, nan=()):
        self.delta, self.nan, self.calls = delta, set(nan), []

    def classify(self, windows, precision=None):
        self.calls.append((precision, len(windows)))
        s = FakeBackend().score(windows)
        if precision in self.nan:
            s = s.copy()
            s[0, 0] = np.nan
        elif precision != "f32":
            s = s + np.float32(self.delta)
        return s


def test_sentinel_windows_are_the_first_candidate_windows_upper_cased_and_padded():
    seq = np.frombuffer(b"acgtNNnn" * 1000 + b"ACGT" * 700, dtype=np.uint8)          # contigs of 8000 and 2800 bases
    off = np.array([0, 8000, 10800], np.int64)
    win = nnc.sentinel_windows(seq, off, False)
    assert win.shape == (2, 6000) and win.dtype == np.uint8                          # 8000 -> one window (tail 2000 < 2500), 2800 -> window 0
    assert bytes(win[0]) == (b"acgtNNnn" * 750).upper() and bytes(win[1][:2800]) == b"ACGT" * 700 and set(win[1][2800:]) == {ord("N")}
    assert len(nnc.sentinel_windows(seq, off, False, limit=1)) == 1
    assert nnc.sentinel_windows(seq, np.array([0], np.int64), False).shape == (0, 6000)


def test_parity_sentinel_measures_logs_and_follows_the_range_fallbacks(monkeypatch):
    logs = []
    console = type("C", (), {"log": lambda self, m, **k: logs.append(m), "error": lambda self, m, **k: logs.append("E " + m)})()
    win = np.frombuffer((b"ACGT" * 1500) * 3, dtype=np.uint8).reshape(3, 6000)
    nnc._WARNED.clear()
    e = _SentinelEngine(delta=2e-5)
    d = nnc.parity_sentinel(e, win, "f16x3tc", console)
    assert abs(d - 2e-5) < 1e-7 and e.calls == [("f32", 3), ("f16x3tc", 3)] and "f16x3tc" in logs[-1] and "2.00e-05" in logs[-1]
    nnc.sentinel_verdict(None, d, console, "f16x3tc")                                 # inside the tolerance: returns
    # non-finite production scores: the sentinel judges the arithmetic that will actually serve the run, and each hop's log line
    # names the arithmetic that FAILED (the second hop used to print the configured one)
    logs.clear()
    e = _SentinelEngine(delta=1e-6, nan=("f16x3tc", "f16x3"))
    d = nnc.parity_sentinel(e, win, "f16x3tc", console)
    assert [c[0] for c in e.calls] == ["f32", "f16x3tc", "f16x3", "bf16x3"] and d < 1e-5
    assert "(f16x3tc)" in logs[0] and "with f16x3." in logs[0] and "(f16x3)" in logs[1] and "with bf16x3." in logs[1] and "of bf16x3 " in logs[2]
    # outside the tolerance: loud exit, status 1, with the way out in the message
    logs.clear()
    d = nnc.parity_sentinel(_SentinelEngine(delta=3e-4), win, "f16x3tc", console)
    with pytest.raises(SystemExit) as ex:
        nnc.sentinel_verdict(None, d, console, "f16x3tc")
    assert ex.value.code == 1 and logs[-1].startswith("E Parity sentinel FAILED") and "GENOMAD_AMD_NO_SENTINEL" in logs[-1]
    # every arithmetic non-finite: inf, fails
    assert nnc.parity_sentinel(_SentinelEngine(nan=("f16x3tc", "f16x3", "bf16x3")), win, "f16x3tc", console) == float("inf")
    # nothing to check: the exact arithmetic itself, no windows, or the opt-out
    assert nnc.parity_sentinel(_SentinelEngine(delta=1.0), win, "f32", console) is None
    assert nnc.parity_sentinel(_SentinelEngine(delta=1.0), win[:0], "f16x3tc", console) is None
    monkeypatch.setenv("GENOMAD_AMD_NO_SENTINEL", "1")
    assert nnc.parity_sentinel(_SentinelEngine(delta=1.0), win, "f16x3tc", console) is None


def test_main_stops_before_writing_anything_when_the_sentinel_trips(tmp_path, monkeypatch, capsys):
    fa = tmp_path / "s.fna"
    _write_fasta(fa, [("c1", "ACGT" * 2000), ("c2", "GGCA" * 1800)])
    monkeypatch.delenv("GENOMAD_AMD_FRONT_END", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(nnc, "_engine", lambda: _SentinelEngine(delta=5e-4))
    with pytest.raises(SystemExit) as ex:
        nnc.main(fa, tmp_path / "bad", False, 128, False, 1, False, False)
    assert ex.value.code == 1 and "Parity sentinel FAILED" in capsys.readouterr().err
    assert not (tmp_path / "bad" / "s_nn_classification").exists()
    log = (tmp_path / "bad" / "s_nn_classification.log").read_text()
    assert "Parity sentinel: max |dscore| of f16x3tc" in log and "5.00e-04" in log
    good = _SentinelEngine(delta=1e-5)
    monkeypatch.setattr(nnc, "_engine", lambda: good)
    nnc.main(fa, tmp_path / "ok", False, 128, False, 1, False, False)
    assert (tmp_path / "ok" / "s_nn_classification" / "s_nn_classification.tsv").exists()
    assert good.calls == [("f32", 2), ("f16x3tc", 2)]                                 # once per run, on the first windows only
    assert "Parity sentinel: max |dscore| of f16x3tc" in (tmp_path / "ok" / "s_nn_classification.log").read_text()
    monkeypatch.setenv("GENOMAD_AMD_NO_SENTINEL", "1")
    monkeypatch.setattr(nnc, "_engine", lambda: _SentinelEngine(delta=5e-4))
    nnc.main(fa, tmp_path / "optout", False, 128, False, 1, False, False)             # opted out: runs through
    assert (tmp_path / "optout" / "s_nn_classification" / "s_nn_classification.tsv").exists()


def _sentinel_rank_worker(rank, world, port, fasta, out_dir, q, bad_rank):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.pop("GENOMAD_AMD_FRONT_END", None)
    from tests.gloo_comm import GlooComm
    nnc._engine = lambda: _SentinelEngine(delta=5e-4 if rank == bad_rank else 1e-6)
    comm = GlooComm(rank, world, port)
    code = 0
    try:
        nnc.main(fasta, out_dir, False, 128, False, 1, False, False, _comm=comm)
    except SystemExit as e:
        code = e.code
    q.put((rank, code))
    comm.close()


def test_sentinel_failure_on_one_rank_stops_every_rank(tmp_path):
    """A rank whose first windows leave the tolerance must not leave the others waiting in the gather: the verdict is collective."""
    mp = pytest.importorskip("torch.multiprocessing")
    import socket
    rng = np.random.default_rng(9)
    fa = tmp_path / "m.fna"
    _write_fasta(fa, [(f"c{i}", "".join(rng.choice(list("ACGT"), 7000))) for i in range(8)])
    for bad_rank, want in ((1, [1, 1]), (-1, [0, 0])):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        out = tmp_path / f"out{bad_rank}"
        procs = [ctx.Process(target=_sentinel_rank_worker, args=(r, 2, port, str(fa), str(out), q, bad_rank)) for r in range(2)]
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=180) for _ in procs)
        for p in procs:
            p.join(timeout=180)
            assert p.exitcode == 0
        assert [g[1] for g in got] == want
        assert (out / "m_nn_classification" / "m_nn_classification.npz").exists() == (bad_rank < 0)


def test_bench_strong_scaling_keeps_the_tuned_launch_shape_at_every_n():
    """at N ranks a step of the job is 65536 / N windows per rank (8192 at N = 8), half the tuned launch.  bench.py
    hands a rank's consecutive steps to the library together so that it launches --chunk windows at every N; the job (K steps), the
    shards and the scores are the same with and without it.  Scaled down 16x over the fake engine: 4096-window steps, chunk 1024."""
    import subprocess
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["GENOMAD_AMD_BENCH_FAKE_ENGINE"] = "1"
    env["OPENBLAS_NUM_THREADS"] = "2"
    got = {}
    for name, extra in (("n8", ["--gpus", "8"]), ("n8_single", ["--gpus", "8", "--no-coalesce"]), ("n1", ["--gpus", "1"])):
        r = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "8", "--warmup", "3", "--windows-per-step", "4096", "--chunk", "1024",
                            "--cpu-sample", "0"] + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[name] = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")][0]
    n8, single, n1 = got["n8"], got["n8_single"], got["n1"]
    assert n8["config"]["windows_per_launch"] == 1024 and n8["config"]["steps_per_call"] == 2          # 2 steps of 512 = one launch of --chunk
    assert n8["roofline"]["launches"] == 4 and n8["roofline"]["flop_per_launch"] == 1024 * 2_762_901_136
    assert single["config"]["windows_per_launch"] == 512 and single["config"]["steps_per_call"] == 1 and single["roofline"]["launches"] == 8
    assert n1["config"]["windows_per_launch"] == 1024 and n1["config"]["steps_per_call"] == 1 and n1["roofline"]["launches"] == 32
    for o in (n8, single, n1):
        assert o["steps"] == 8 and o["parity"]["windows"] == 8 * 4096 and o["steps_verified"]["mismatching_windows_all_ranks"] == 0
        assert o["max_abs_dscore"] == 0.0 and "failed" not in o


def test_hbm_traffic_json_is_a_function_of_the_committed_pmc_passes():
    """bench.py scales `roofline.traffic` from profiles/hbm_traffic.json; the file must be reproducible, to the
    digit, from the pmc_1.txt / pmc_2.txt summaries committed under profiles/ (scripts/hbm_traffic_json.py --check)."""
    import subprocess
    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "scripts" / "hbm_traffic_json.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    import bench
    per_window, source = bench.hbm_traffic("f16x3tc")
    assert per_window > 6012 and "pmc_1.txt" in source


def test_native_accession_digests_equal_the_python_mirror():
    """the sharded validation hashes accessions in ONE native pass (gnn_fasta_accession_digests).  Same records (every
    record with a non-empty raw sequence, all-N ones included), same accession rule as header.split()[0] (genomad/sequence.py:24-25)
    - and a header whose first token has a non-ASCII byte (Python knows non-ASCII white space) or a text with '\\r' takes the Python
    route and still gives the digests the other ranks would compute natively."""
    def arr(b):
        return np.frombuffer(bytearray(b), dtype=np.uint8)

    text = (b"junk before\n>c1 desc\nACGT\nAC\n>  lead\tx\nNNNN\n>empty_seq\n>tab\tsep\nA\n>fs\x1cx\nG\n>c1\nT\n>last")
    want_acc = ["c1", "lead", "tab", "fs", "c1"]                    # ">empty_seq" and ">last" have no sequence: not counted (index mode)
    assert sequence.index_accessions(arr(text)) == want_acc
    got = sequence.accession_digests_of_text(arr(text))
    assert got.dtype == np.dtype("<u8") and list(got) == [sequence._digest_of_accession(a) for a in want_acc]
    assert got[0] == got[4] and len(set(got.tolist())) == 4
    assert list(sharding.accession_digests(want_acc)) == list(got)
    # non-ASCII white space inside the first token: Python splits there, bytes >= 0x80 send the piece down the Python route
    nb = ">a b rest\nACGT\n>café x\nAC\n".encode()
    assert sequence.index_accessions(arr(nb)) == ["a", "café"]
    assert list(sequence.accession_digests_of_text(arr(nb))) == [sequence._digest_of_accession("a"), sequence._digest_of_accession("café")]
    # universal newlines
    cr = b">x y\r\nAC\r\n>z\rGG\r"
    assert list(sequence.accession_digests_of_text(arr(cr))) == [sequence._digest_of_accession("x"), sequence._digest_of_accession("z")]
    assert len(sequence.accession_digests_of_text(arr(b""))) == 0 and len(sequence.accession_digests_of_text(arr(b"no records\n"))) == 0
    # fasta_verdict takes the digests as they are; equal digests inside a share go to the exact check
    calls = []
    assert sharding.fasta_verdict(None, got[:4], lambda: calls.append(1) or True) and not calls
    assert not sharding.fasta_verdict(None, got, lambda: calls.append(1) or False) and calls == [1]
    assert not sharding.fasta_verdict(None, got[:0], lambda: True)


def test_cpu_baseline_is_not_throttled_by_the_launcher_and_keeps_its_budget():
    """`python -m torch.distributed.run --nproc-per-node N` (N > 1) exports OMP_NUM_THREADS=1, which OpenBLAS honours: rank 0's
    1 024-window CPU baseline would run on ONE thread (~25 minutes) while the other ranks wait for its RCCL id - the driver's
    scaling run would be the first to notice.  bench.cpu_baseline raises the BLAS pool to the cores the process may use whatever
    the environment says, and cuts the sample to its time budget."""
    import subprocess
    root = Path(__file__).resolve().parents[1]
    code = ("import sys, json; sys.path.insert(0, %r); import bench; from genomad_amd import synthetic\n"
            "b, s = bench.cpu_baseline(synthetic.synth_weights(), 8, batch=2, budget_s=%s)\n"
            "print(json.dumps({'cores': b['cores'], 'env': b['omp_num_threads_env'], 'n': len(s), 'sample': b['sample'], 'visible': b['host_cpus_visible']}))")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-c", code % (str(root), "600")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    o = json.loads(r.stdout.strip().splitlines()[-1])
    assert o["env"] == "1" and o["n"] == 8
    assert o["cores"] == o["visible"] or o["visible"] == 1, o           # not the launcher's single thread
    r = subprocess.run([sys.executable, "-c", code % (str(root), "0.001")], env=env, capture_output=True, text=True, timeout=600)
    o = json.loads(r.stdout.strip().splitlines()[-1])
    assert o["n"] == 2 and "cut from 8" in o["sample"]                   # one batch, then the budget is gone: said in the line


def test_bench_under_the_drivers_launcher_with_two_ranks_over_the_fake_engine():
    """The driver's N > 1 command verbatim - `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` - on the CPU over tests/fake_engine.py: the launcher's environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT, and its OMP_NUM_THREADS=1) reaches bench.py, the ranks find each other through
    the rendezvous file keyed on the launcher's identity, rank 0 prints ONE JSON line with a CPU baseline that was not throttled to
    one thread, and every rank exits 0."""
    import socket
    import subprocess
    pytest.importorskip("torch")
    root = Path(__file__).resolve().parents[1]
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "OMP_NUM_THREADS")}
    env["GENOMAD_AMD_BENCH_FAKE_ENGINE"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--windows-per-step", "512", "--cpu-sample", "4"], env=env, capture_output=True, text=True, timeout=900, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout
    o = lines[0]
    assert o["n_gpus"] == 2 and o["rccl_ranks"] == 2 and o["steps"] == 4 and o["fake_engine"] is True
    assert o["cpu_baseline"]["omp_num_threads_env"] == "1"                      # the launcher did export it ...
    assert o["cpu_baseline"]["cores"] == o["cpu_baseline"]["host_cpus_visible"] or o["cpu_baseline"]["host_cpus_visible"] == 1   # ... and it did not bind
    assert o["steps_verified"]["mismatching_windows_all_ranks"] == 0 and o["parity"]["ok"] and "failed" not in o


# ------------------------------------------------------------------ f16x3tk: when main() builds / uses the k-mer tables (round 6)
class _TablesEngine:
    """stand-in for NNEngine's three table methods"""

    def __init__(self, fits=True, has=False):
        self.fits, self.has, self.builds = fits, has, 0

    def has_kmer_tables(self):
        return self.has

    def build_kmer_tables(self, reserve_bytes=-1):
        self.builds += 1
        self.has = self.has or self.fits
        return self.has


class _Comm:
    """allgather_i64 over a fixed list of what the OTHER rank contributes to each collective"""

    def __init__(self, others):
        self.others = list(others)

    def allgather_i64(self, values):
        return np.array([list(values), [int(self.others.pop(0))]], dtype=np.int64)


def test_select_arithmetic_policy(monkeypatch):
    """nn_classification.select_arithmetic: the default arithmetic is promoted to f16x3tk when the engine holds the tables, when the
    input is large enough to repay their set-up (GENOMAD_AMD_KMER_TABLES_MIN_GB) or when GENOMAD_AMD_KMER_TABLES=1; never under
    =0; a device that cannot hold them keeps f16x3tc; an explicit f16x3tk that cannot be served is an error; other arithmetics pass
    through; with several ranks ONE rank that cannot build keeps every rank on the default (scores must not depend on the rank)."""
    from genomad_amd import nn_classification as nnc
    for k in ("GENOMAD_AMD_KMER_TABLES", "GENOMAD_AMD_KMER_TABLES_MIN_GB"):
        monkeypatch.delenv(k, raising=False)
    small, big = 10 ** 9, 30 * 10 ** 9
    e = _TablesEngine()
    assert nnc.select_arithmetic(e, "f16x3tc", small) == "f16x3tc" and e.builds == 0
    assert nnc.select_arithmetic(e, "f16x3tc", big) == "f16x3tk" and e.builds == 1
    assert nnc.select_arithmetic(e, "f16x3tc", small) == "f16x3tk"              # the engine holds them now
    for other in ("bf16x3", "f32", "f16x3"):
        assert nnc.select_arithmetic(_TablesEngine(), other, big) == other
    assert nnc.select_arithmetic(_TablesEngine(fits=False), "f16x3tc", big) == "f16x3tc"
    with pytest.raises(RuntimeError, match="cannot hold"):
        nnc.select_arithmetic(_TablesEngine(fits=False), "f16x3tk", small)
    assert nnc.select_arithmetic(_TablesEngine(), "f16x3tk", small) == "f16x3tk"
    monkeypatch.setenv("GENOMAD_AMD_KMER_TABLES", "0")
    assert nnc.select_arithmetic(_TablesEngine(has=True), "f16x3tc", big) == "f16x3tc"
    monkeypatch.setenv("GENOMAD_AMD_KMER_TABLES", "1")
    assert nnc.select_arithmetic(_TablesEngine(), "f16x3tc", small) == "f16x3tk"
    monkeypatch.setenv("GENOMAD_AMD_KMER_TABLES", "sometimes")
    with pytest.raises(ValueError, match="GENOMAD_AMD_KMER_TABLES"):
        nnc.select_arithmetic(_TablesEngine(), "f16x3tc", small)
    monkeypatch.setenv("GENOMAD_AMD_KMER_TABLES", "auto")
    monkeypatch.setenv("GENOMAD_AMD_KMER_TABLES_MIN_GB", "0.5")
    assert nnc.select_arithmetic(_TablesEngine(), "f16x3tc", small) == "f16x3tk"
    monkeypatch.delenv("GENOMAD_AMD_KMER_TABLES_MIN_GB")
    # several ranks: first collective = who wants them, second = who could build them
    assert nnc.select_arithmetic(_TablesEngine(), "f16x3tc", big, _Comm([1, 1])) == "f16x3tk"
    assert nnc.select_arithmetic(_TablesEngine(), "f16x3tc", big, _Comm([1, 0])) == "f16x3tc"       # the other rank could not build
    e = _TablesEngine()
    assert nnc.select_arithmetic(e, "f16x3tc", small, _Comm([1, 1])) == "f16x3tk" and e.builds == 1   # the other rank already holds them
    e = _TablesEngine()
    assert nnc.select_arithmetic(e, "f16x3tc", small, _Comm([0, 1])) == "f16x3tc" and e.builds == 0   # nobody wants them
    # an engine without the table methods (the CPU stand-in of the multi-rank tests) is left alone
    assert nnc.select_arithmetic(object(), "f16x3tc", big) == "f16x3tc"
    assert nnc.RANGE_FALLBACKS["f16x3tk"] == ("f16x3", "bf16x3")
