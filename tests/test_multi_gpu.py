"""Multi-GPU execution as TESTS: RCCL with more than one rank has never run on the one-GPU boxes this repo
is built on, so the first multi-rank execution must not be a bench.  The `-m gpu` tests here skip when fewer than two devices
are visible and otherwise run D = min(devices, 8) ranks, one process per GPU, started the way bench.py starts its own ranks
(bench.spawn_ranks: RANK / LOCAL_RANK / WORLD_SIZE + a private rendezvous directory):

  (i)   every collective of genomad_amd/rccl.py (ncclAllGather, ncclGather host and device, all-reduce max, barrier) across D devices;
  (ii)  `bench.py --gpus D --steps 2 --windows-per-step 4096 D`: exit 0, rccl_ranks == D, no mismatching window, and the gathered
        scores bit-equal to a 1-rank run of the same job (sharding the reference's predict loop, nn_classification.py:316-320,
        must not change a bit);
  (iii) main() under D ranks on a small FASTA == single process, bit for bit (contigs sharded, ONE gather, rank 0 writes).

The CPU variants run the SAME worker and the same assertions over tests/fake_engine.py (GENOMAD_AMD_BENCH_FAKE_ENGINE=1: a fake
comm library exchanging through files) and, for main(), over gloo (tests/test_host.py)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
WORKER = str(ROOT / "tests" / "multi_gpu_worker.py")


def _devices() -> int:
    import ctypes
    from genomad_amd import _lib
    try:
        n = ctypes.c_int()
        if _lib.load().gnn_device_count(ctypes.byref(n)) != 0:
            return 0
        return n.value
    except Exception:  # noqa: BLE001
        return 0


def _clean_env(fake: bool):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GENOMAD_AMD_BENCH_FAKE_ENGINE")}
    if fake:
        env["GENOMAD_AMD_BENCH_FAKE_ENGINE"] = "1"
    env["OPENBLAS_NUM_THREADS"] = "2"
    return env


def _spawn(world, argv, env):
    """bench.spawn_ranks in a child interpreter (it inherits stdout for rank 0; here everything is captured)."""
    code = ("import sys, json; sys.path.insert(0, %r); import bench; sys.exit(bench.spawn_ranks(%d, %r))" % (str(ROOT), world, [sys.executable] + argv))
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)


def _check_comm(world, tmp_path, fake):
    out = tmp_path / f"comm_{world}.json"
    r = _spawn(world, [WORKER, "comm", str(out)], _clean_env(fake))
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(out))
    assert res["world"] == world
    for k in ("allgather_ok", "gather_root0_ok", "gather_dev_ok", "allreduce_max_ok", "all_ranks_ok"):
        assert res[k] is True, (k, res)
    return res


def _check_bench(world, tmp_path, fake):
    env = _clean_env(fake)
    lines = {}
    for n in (world, 1):
        dump = tmp_path / f"scores_{n}.npy"
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--windows-per-step",
                            str(4096 * world), "--cpu-sample", "0", "--dump-scores", str(dump)], env=env, capture_output=True, text=True, timeout=1800)
        assert r.returncode == 0, r.stderr[-3000:]
        lines[n] = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")][0]
    o = lines[world]
    assert o["n_gpus"] == world and o["rccl_ranks"] == world and o["scaling"] == "strong" and o["steps"] == 2
    assert o["steps_verified"]["mismatching_windows_all_ranks"] == 0 and o["steps_verified"]["windows_all_ranks"] == 2 * 4096 * world
    assert o["parity"]["ok"] and o["parity"]["windows"] == 2 * 4096 * world and "failed" not in o
    assert len(o["per_rank_windows_per_s"]) == world and o["gather_ms"] > 0
    many, one = np.load(tmp_path / f"scores_{world}.npy"), np.load(tmp_path / "scores_1.npy")
    assert many.shape == one.shape == (2 * 4096 * world, 3) and many.dtype == np.float32
    assert np.array_equal(many, one), "sharding over ranks changed scores"
    return o, lines[1]


# ------------------------------------------------------------------ CPU: the same checks over the fake engine / fake comm library
@pytest.mark.parametrize("world", [2, 3])
def test_comm_collectives_over_the_fake_comm_library(tmp_path, world):
    _check_comm(world, tmp_path, fake=True)


def test_bench_multi_rank_scores_equal_one_rank_over_the_fake_engine(tmp_path):
    _check_bench(2, tmp_path, fake=True)


# ------------------------------------------------------------------ GPU: D real devices, RCCL over xGMI
def _world_or_skip():
    d = _devices()
    if d < 2:
        pytest.skip(f"{d} GPU(s) visible: the multi-GPU tests need at least 2 (they run on any multi-GPU MI355X box)")
    return min(d, 8)


@pytest.mark.gpu
def test_rccl_collectives_across_all_devices(tmp_path):
    world = _world_or_skip()
    res = _check_comm(world, tmp_path, fake=False)
    assert res["devices"] >= world
    print(f"RCCL: allgather / gather (host, device) / allreduce max / barrier across {world} devices ok")


@pytest.mark.gpu
def test_bench_multi_rank_scores_equal_one_rank(tmp_path):
    world = _world_or_skip()
    o, one = _check_bench(world, tmp_path, fake=False)
    print(f"bench.py --gpus {world}: {o['value']:.0f} windows/s ({one['value']:.0f} on one), gather {o['gather_ms']:.2f} ms "
          f"(isolated {o['gather_ms_isolated']:.3f} ms), comm init {o['comm_init_s']:.2f} s")


@pytest.mark.gpu
def test_main_under_all_devices_equals_single_process(tmp_path, synth_weights):
    world = _world_or_skip()
    from genomad_amd import weights as W
    rng = np.random.default_rng(11)
    fa = tmp_path / "multi.fna"
    with open(fa, "w") as f:
        for i, L in enumerate(rng.integers(2000, 40000, 6 * world)):
            f.write(f">contig_{i} len={L}\n{''.join(rng.choice(list('ACGTN'), int(L), p=[.24, .24, .24, .24, .04]))}\n")
    wpath = tmp_path / "w.npz"
    W.save_npz(wpath, synth_weights)
    env = _clean_env(False)
    env["GENOMAD_AMD_WEIGHTS"] = str(wpath)
    env.pop("GENOMAD_AMD_FRONT_END", None)
    r = _spawn(world, [WORKER, "main", str(fa), str(tmp_path / "many")], env)
    assert r.returncode == 0, r.stderr[-3000:]
    r1 = subprocess.run([sys.executable, WORKER, "main", str(fa), str(tmp_path / "one")], env=env, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-3000:]
    a = np.load(tmp_path / "many" / "multi_nn_classification" / "multi_nn_classification.npz")
    b = np.load(tmp_path / "one" / "multi_nn_classification" / "multi_nn_classification.npz")
    assert list(a["contig_names"]) == list(b["contig_names"]) and len(a["contig_names"]) == 6 * world
    assert np.array_equal(a["predictions"], b["predictions"])
    ta = (tmp_path / "many" / "multi_nn_classification" / "multi_nn_classification.tsv").read_bytes()
    assert ta == (tmp_path / "one" / "multi_nn_classification" / "multi_nn_classification.tsv").read_bytes()
