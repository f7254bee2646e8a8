#!/usr/bin/env python
"""bench.py — 6 kbp windows classified/sec on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  Under a launcher (torch.distributed.run is only that: RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_PORT are read from the environment) every process is one rank; a PLAIN `python bench.py --gpus N` with
N > 1 spawns its N ranks itself (spawn_ranks) and rank 0's JSON line is the output.  Nothing here imports
torch: device memory, streams and events come from the C ABI of libgenomad_nn_hip.so; the barrier, the
max-over-ranks and the final gather are RCCL through its gnn_comm_* entry points (genomad_amd/rccl.py) — at
N = 1 too: the communicator is always created, so the N = 1 line goes through the same ncclGather as N = 8.

Workload (config.workload): synthetic 6 kbp windows (seed 1234) + synthetic weights (seed 42) of the
reference shapes, resident in HBM before the timed region.  A "step" is one pass of the whole hot path
(tokenise -> conv1..3 -> IGLOO heads -> dense/softmax) over one batch of --windows-per-step windows.
  --scaling strong (default, BASELINE configs[2]/[3]: "1 M windows sharded across the GPUs"): the K x 65536 =
      1 M windows of the job (default K = 16) are split into contiguous shards, rank r classifies windows
      [r*n/N, (r+1)*n/N) (every step = 65536 / N windows per rank, in launches of --chunk), ONE RCCL gather of the
      scores to rank 0 at the end.
  --scaling weak: every rank classifies its own K x 65536 windows.
  --workload metagenome --gbp-total G (BASELINE configs[4]): a synthetic metagenome of G Gbp (contigs of
      1-500 kbp, log-uniform) sharded by contigs over the ranks and streamed through HBM in chunks of
      --gbp-per-step; spans -> N rule -> encode+IGLOO -> per-contig mean on the device, per-contig scores
      gathered on rank 0.
`value` = windows of all ranks / max-over-ranks wall time, barrier + stream sync on both sides, the gather
and the copy of the scores to the host of rank 0 inside the timed region.

Arithmetic (--precision auto, the default): "f16x3tk" when EVERY rank's device can hold the 156 GB of k-mer tables (conv2 read from a
table of all 14-mers, head A from 9-mer tables: genomad_amd/csrc/gnn_fused_tk.hip) - built by gnn_build_kmer_tables before anything is
timed, `kmer_tables` in the line says how long that took - else the default "f16x3tc"; --no-kmer-tables measures the latter.
`roofline.table_reads` lists what f16x3tk reads instead of computing (half of the algorithmic FLOPs) and the bytes that costs.

After the timed region (untimed) every window of the job is classified once more with the exact-f32 device path and,
bit for bit, through the synchronous entry point: `parity` reports max |dscore| over ALL timed windows, how many exceed
1e-4 and 5e-5, and the process exits non-zero if any window exceeds the 1e-4 tolerance or any bit differs.

At N = 1 the default line also carries two untimed blocks (<= 25 s together, never part of `value`):
  main_e2e      the drop-in entry point genomad_amd.nn_classification.main() with the reference's signature
                (nn_classification.py:21-30) on a config-1-shaped FASTA (8 records, ~5.4 Mbp, 906 windows; TSV rows of the seven
                plasmids against the oracle chain formatted like :344-348) and on a ~600 MB synthetic FASTA written to the temp
                directory: seconds, windows/s, the parity sentinel's value
  metagenome    3 Gbp of BASELINE configs[4] through the contig front end: windows/s, and one 48 Mbp chunk bit-equal to the
                host rules + the window path
A failure of any rank (exception, signal from the launcher, a stage that runs into its time-out) prints ONE JSON line with "error",
the rank, the stage it was in and the last library error, and exits non-zero.

Extra objects in the JSON line:
  roofline      dominant kernel (fused front end): algorithmic FLOP per launch / HIP-event duration measured
                live on the library's stream, against the dense bf16/f16 MFMA peak (2.5 PFLOP/s)
  cpu_baseline  the numpy restatement of the reference (oracle/, reference-faithful mode: explicit one-hot,
                dense conv1, batch 128 like nn_classification.py's default) timed on the host cores on a
                bounded sample; N=1, rank 0 only
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_WINDOW = 2_762_901_136        # conv2+conv3, w_v x2, IGLOO small terms, head
REF_FLOP_PER_WINDOW = FLOP_PER_WINDOW + 2 * 5997 * 6 * 257 * 128   # + conv1 as the dense one-hot contraction TF runs
MFMA_PEAK_TFLOPS = 2500.0              # MI355X_MICROARCH.md: dense bf16 / f16 MFMA peak
ENCODER_BYTES = {"u8": 6000 + 5997 * 257, "bf16": 6000 + 5997 * 257 * 2, "f32": 6000 + 5997 * 257 * 4}
HBM_PEAK_GBS = 8000.0
# matrix-pipe cost of one product in units of one bf16/f16 pass
# f16x3tc: conv2 + conv3 (85.35 % of the algorithmic FLOPs) issue 8 / 18 of the direct form's MFMAs, y @ w_v and the rest all of them
# since round 6 head A's y @ w_v (7.11 %) is a table lookup: no MFMAs at all
# f16x3tk (round 6): conv2 (42.675 %) and head A's pair products are table reads as well: conv3 by Toom-Cook and head B's y @ w_v are what is left
MFMA_PASSES = {"f16c6": 1.5, "f16x3": 3.0, "f16x3tc": round(3.0 * (0.8535 * 8 / 18 + 0.1465 - 0.0711), 3), "bf16x3": 3.0,
               "f16x3tk": round(3.0 * (0.42675 * 8 / 18 + 0.1465 - 0.0711), 3)}
# the dominant kernel as rocprofv3's kernel trace names it (profiles/*/kernel_stats.csv)
FRONT_KERNEL = {"f16x3tk": "gnn::tk::fused_front_tk_kernel<false>", "f16x3tc": "gnn::tc::fused_front_tc_kernel<false>", "f16x3": "gnn::x3::fused_front_x3_kernel<true, false>", "bf16x3": "gnn::x3::fused_front_x3_kernel<false, false>",
                "f16c6": "gnn::c6::fused_front_c6_kernel", "f32": "f32 front end (5 kernels)"}
DTYPE_TEXT = {"f16c6": "f16 MFMA + MX-fp6 (e2m3, both operands block scaled) correction MFMAs, f32 accumulate (1.5 f16-pass equivalents)",
              "f16x3": "f16x3 (split-f16 MFMA, 3 passes, f32 accumulate; logits GEMM split-f16 x 3 as well, dense head exact f32)",
              "f16x3tc": "f16x3tc (split-f16 MFMA, 3 products per operand pair, f32 accumulate; conv2 / conv3 by Toom-Cook F(3,6) minimal filtering over "
                         "time with f32 transforms: 0.444x their MFMAs; head A's y @ w_v rows gathered from a 9-mer table (exact f32), head B's "
                         "direct; logits GEMM split-f16 x 3, dense head exact f32)",
              "f16x3tk": "f16x3tk (conv2 read from a 137 GB table of all 14-mers in HBM - x2[t] is a function of the bases t-10 .. t+3 -, head A's pair "
                         "products from an (entry, 9-mer) table, head A's y @ w_v rows from a 9-mer table, all exact f32; conv3 by Toom-Cook F(3,6) "
                         "with split-f16 MFMA, 3 products per operand pair, f32 accumulate; head B's y @ w_v direct; logits GEMM split-f16 x 3, "
                         "dense head exact f32)",
              "bf16x3": "bf16x3 (split-bf16 MFMA, 3 passes, f32 accumulate)", "f32": "f32"}


def cpu_baseline(weights, sample: int, batch: int = 128, budget_s: float = 40.0):
    """Time the CPU restatement (the checker) on up to `sample` windows; return (dict, scores of the windows it classified).

    Two guards for launchers and slow hosts (found in round 5: `python -m torch.distributed.run --nproc-per-node N` with N > 1 exports
    OMP_NUM_THREADS=1, which OpenBLAS honours - the 1 024-window sample would then take ~25 minutes on rank 0 while the other ranks
    wait for its RCCL id): the BLAS pool is raised to the cores this process may run on for the duration of the baseline, whatever
    the environment says, and the sample is cut to what fits `budget_s` seconds after the first timed batch."""
    import contextlib
    import numpy as np
    from genomad_amd import synthetic
    from oracle import igloo_oracle, sequence_oracle
    cores = len(os.sched_getaffinity(0))
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        pool = threadpool_limits(limits=cores)
    except Exception:  # noqa: BLE001
        threadpool_info, pool = None, contextlib.nullcontext()
    with pool:
        blas_desc = None
        try:
            pools = [p_ for p_ in threadpool_info() if p_.get("user_api") == "blas"] or threadpool_info()
            blas_threads = max([p_.get("num_threads", 1) for p_ in pools] or [1])
            top = max(pools, key=lambda p_: p_.get("num_threads", 1)) if pools else {}
            blas_desc = {k: top.get(k) for k in ("internal_api", "version", "threading_layer", "architecture") if top.get(k) is not None}
        except Exception:  # noqa: BLE001
            blas_threads = cores
        bases = synthetic.synth_windows(0, sample)
        tok = sequence_oracle.tokenize_closed_form(bases)
        igloo_oracle.forward(tok[:2], weights, np.float32, dense_onehot=True, shifted_conv=True)      # warm up BLAS
        t = time.perf_counter()
        scores, done = [], 0
        for a in range(0, sample, batch):
            scores.append(igloo_oracle.forward(tok[a:a + batch], weights, np.float32, dense_onehot=True, shifted_conv=True))
            done = min(a + batch, sample)
            per_batch = (time.perf_counter() - t) / (a // batch + 1)
            if (time.perf_counter() - t) + per_batch > budget_s and done < sample:
                break                                     # the next batch would leave the budget: a smaller sample, said below
        dt = time.perf_counter() - t
        m = min(batch, done)
        t2 = time.perf_counter()
        igloo_oracle.forward(tok[:m], weights, np.float32, dense_onehot=False, shifted_conv=True, literal=False)
        dt_alg = (time.perf_counter() - t2) / m
    cut = "" if done == sample else f" (cut from {sample} to stay inside {budget_s:.0f} s)"
    return ({"value": round(done / dt, 2), "unit": "windows/s", "cores": int(blas_threads), "kind": "port",
             "sample": f"{done} synthetic windows{cut} (the first of the GPU workload), numpy fp32 restatement of the "
                       f"reference in reference-faithful mode (explicit 5997x257 one-hot, dense conv1, IGLOO "
                       f"kernel op for op), batch {batch} (the reference's default batch size), OpenBLAS "
                       f"threads={blas_threads}: {done / dt * REF_FLOP_PER_WINDOW / 1e9:.0f} GFLOP/s of the "
                       f"{REF_FLOP_PER_WINDOW / 1e9:.2f} GFLOP/window this mode executes; algorithmic mode "
                       f"(conv1 as gather, closed-form IGLOO): {1.0 / dt_alg:.1f} windows/s. TensorFlow itself is "
                       f"not installable here. Baseline, not target.",
             "gflops": round(done / dt * REF_FLOP_PER_WINDOW / 1e9, 1), "seconds": round(dt, 1),
             "host_cpus_visible": cores, "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS"), "blas": blas_desc,
             # `cores` = the threads the BLAS pool actually ran with.  The pool is asked for every core this process may run on
             # (threadpool_limits(limits=host_cpus_visible)); a smaller number is the library's own ceiling, not a choice
             "threads_cap_reason": None if blas_threads >= cores else
             f"threadpool_limits(limits={cores}) was requested; the BLAS build in this image grants at most {blas_threads} threads "
             f"(OpenBLAS' compile-time NUM_THREADS), so {cores - blas_threads} of the {cores} visible cores stay idle during the baseline"},
            np.concatenate(scores))


def encoder_block(eng, rank: int, launches: int = 8, windows: int = 2048):
    """The stand-alone byte -> one-hot encoder (model.py:9-11) for the three output dtypes: windows/s and
    HBM GB/s of `onehot_kernel` from HIP events on the library's stream (algorithmic bytes = 6000 in + 5997 x 257 x itemsize out per
    window: the kernel's WRITE_SIZE is 1.000 x that, profiles/hbm_traffic.json), against the 8 TB/s HBM3E peak.  Untimed part of the
    classification line (< 0.2 s of GPU time); `bench.py --kernel encoder` is the stand-alone bench of the same kernel."""
    from genomad_amd import _lib
    free_b = eng.mem_info()[0] if hasattr(eng, "mem_info") else None
    while free_b is not None and windows > 64 and windows * 5997 * 257 * 4 > free_b // 2:
        windows //= 2                                     # 12.6 GB of f32 one-hot at 2048 windows: halve on a short device
    bases = eng.alloc(windows * 6000)
    try:
        out = eng.alloc(windows * 5997 * 257 * 4)
    except Exception:
        bases.free()
        raise
    res = {"kernel": "gnn::onehot_kernel", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "windows_per_launch": windows,
           "launches": launches, "frac": {}, "achieved": {}, "windows_per_s": {}, "avg_launch_ms": {}, "bytes_per_window": dict(ENCODER_BYTES)}
    try:
        eng.synth_windows_dev(rank * windows, windows, bases.ptr)
        for name, code in (("u8", _lib.OH_U8), ("bf16", _lib.OH_BF16), ("f32", _lib.OH_F32)):
            for _ in range(2):
                _lib.check(eng.lib.gnn_onehot_dev(eng.ctx, bases.ptr, windows, code, out.ptr))
            eng.profile_enable(True)
            eng.profile_reset()
            for _ in range(launches):
                _lib.check(eng.lib.gnn_onehot_dev(eng.ctx, bases.ptr, windows, code, out.ptr))
            ms, n = eng.profile_get(_lib.K_ENCODER)
            eng.profile_enable(False)
            avg = ms / max(n, 1)
            gbs = ENCODER_BYTES[name] * windows / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
            res["achieved"][name] = round(gbs, 1)
            res["frac"][name] = round(gbs / HBM_PEAK_GBS, 4)
            res["windows_per_s"][name] = round(windows / (avg * 1e-3), 1) if avg > 0 else 0.0
            res["avg_launch_ms"][name] = round(avg, 5)
    finally:
        bases.free()
        out.free()
    return res


def hbm_traffic(precision: str):
    """(bytes per window, source text) of the dominant kernel's HBM-side traffic from the committed PMC passes
    (profiles/hbm_traffic.json, regenerated from profiles/r05/rocprof/pmc_1.txt and pmc_2.txt by scripts/hbm_traffic_json.py)."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(tpath):
        return None, None
    tj = json.load(open(tpath))
    per_window = tj.get("bytes_per_window", {}).get(precision)
    if per_window is None:
        return None, None
    return per_window, ("NOT measured by this run: 2*FETCH_SIZE + WRITE_SIZE per window from separate rocprofv3 --pmc passes of this "
                        "command (" + tj.get("source", "profiles/README.md") + "), scaled to this run's windows per launch")


def front_roofline(precision, windows_per_launch, avg_ms, launches, front_ms, back_ms):
    """The `roofline` object of the dominant kernel (fused front end): algorithmic FLOP per launch / mean HIP-event launch duration."""
    tflops = FLOP_PER_WINDOW * windows_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    per_window, source = hbm_traffic(precision)
    traffic = int(per_window * windows_per_launch) if per_window is not None else None
    alg = int(6012 * windows_per_launch)
    tk = precision == "f16x3tk"
    # f16x3tk: the table rows a window reads ARE its algorithm's bytes: 63 steps x 101 rows x 512 B of the 14-mer table (the five carry rows
    # of a step are read again), 5 992 rows x 512 B of head A's y @ w_v table, 8 400 4-byte entries of head A's pair-product table
    tk_bytes = {"x2_rows_14mer_table": 63 * 101 * 512, "wva_rows_9mer_table": 5992 * 512, "pair_products_a": 8400 * 4, "bases": 6000, "scores": 12}
    if tk:
        alg = int(sum(tk_bytes.values()) * windows_per_launch)
    elif precision == "f16x3tc":                       # since round 6 head A's y @ w_v rows are read from the 9-mer table: 5 992 rows x 512 B per window
        alg = int((6012 + tk_bytes["wva_rows_9mer_table"]) * windows_per_launch)
    r = {
        "bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(tflops / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": source,
        "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": round(traffic / alg, 1) if traffic and alg else None,
        "traffic_note": "HBM is not the bound; algorithmic bytes = bases + scores (6 012 B) + for f16x3tc the 5 992 rows of head A's 9-mer table a window "
                        "gathers (3.07 MB); the excess is the f32 spill of the pooled y @ w_v rows (767 KB/window) and pair products (67 KB/window) from "
                        "the front end to the back end, written once and read once (the softmax over the 749 pooled positions needs all of a window's "
                        "2 100 patches first, DESIGN.md section 3), and for f16x3tc the conv1 pair tables and folded IGLOO weights that miss the L2",
        "kernel": FRONT_KERNEL[precision],
        "flop_per_launch": int(FLOP_PER_WINDOW * windows_per_launch), "avg_launch_ms": round(avg_ms, 4),
        "launches": int(launches), "mfma_passes": MFMA_PASSES.get(precision),
        "note": "achieved counts ALGORITHMIC flops (2.763 GFLOP/window) against the dense 16-bit MFMA peak; the "
                "1e-4 tolerance needs more than one 16-bit pass per product (profiles/history/r02_precision_study.json): "
                "mfma_passes bf16-pass equivalents are issued, which caps frac at 1/mfma_passes",
        "backend_ms_total": round(back_ms, 2), "front_ms_total": round(front_ms, 2)}
    if tk:
        share = 0.42675 + 0.0711                       # conv2 + head A's y @ w_v of the algorithmic FLOPs are table reads
        r["table_reads"] = {
            "share_of_algorithmic_flops": round(share, 4), "flop_executed_per_window": int(FLOP_PER_WINDOW * (1.0 - share)),
            "frac_counting_executed_flops_only": round(tflops * (1.0 - share) / MFMA_PEAK_TFLOPS, 4),
            "bytes_per_window": tk_bytes, "gather_gb_per_s": round(sum(tk_bytes.values()) * windows_per_launch / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else None,
            "gather_frac_of_hbm_peak": round(sum(tk_bytes.values()) * windows_per_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if avg_ms > 0 else None,
            "probe": "scripts/probe_gather_big.hip: 512-byte rows at random 14-mers of a 137 GB table arrive at 5.9 TB/s (11.6 G rows/s) on this chip: "
                     "the gather is not the bound, the conv3 weight stream (L2 -> CU) and the board's power are"}
        r["note"] = ("achieved counts ALGORITHMIC flops (2.763 GFLOP/window) against the dense 16-bit MFMA peak; f16x3tk EXECUTES half of them (conv3 by "
                     "Toom-Cook, head B's y @ w_v: mfma_passes bf16-pass equivalents per algorithmic FLOP) and READS the other half - conv2 and head A - "
                     "from k-mer tables in HBM (table_reads): x2[t] is a function of 14 bases, 4^14 rows of 512 B fit one MI355X's 288 GB")
        r["traffic_note"] = ("algorithmic bytes = the table rows a window gathers (table_reads.bytes_per_window: 6.4 MB) + bases + scores; traffic = "
                             "fabric-side counters (Infinity-Cache hits included) of the PMC passes named in traffic_source; the excess over the "
                             "algorithmic bytes is the f32 spill of the pooled y @ w_v rows (767 KB/window) and pair products (67 KB/window) to the back end")
    return r


class PowerSampler:
    """`rocm-smi -d <gpu> --showpower` every ~0.4 s on a helper thread (a separate process: nothing is enqueued on the GPU)."""

    def __init__(self, device: int):
        import re
        import subprocess
        import threading
        self.samples, self._stop = [], threading.Event()

        def run():
            while not self._stop.is_set():
                try:
                    out = subprocess.run(["rocm-smi", "-d", str(device), "--showpower"], capture_output=True, text=True, timeout=10).stdout
                    m = re.search(r"Power \(W\):\s*([0-9.]+)", out)
                    if m:
                        self.samples.append(float(m.group(1)))
                except Exception:  # noqa: BLE001
                    pass
                self._stop.wait(0.4)
        self._t = threading.Thread(target=run, daemon=True)
        self._t.start()

    def stop(self):
        self._stop.set()
        self._t.join(timeout=15)
        return self.samples


# ---------------------------------------------------------------------------------------------------------------------
# Failure reporting.  Every rank keeps the name of the stage it is in; whatever ends a rank early - an
# exception, the launcher's SIGTERM after another rank died, a stage that exceeds its time-out - goes through report_failure():
# the FIRST rank of the launch to get there (an exclusive marker file in the launch's rendezvous directory) prints ONE JSON
# line with "error", its rank, the stage, the RCCL ranks it has seen and the library's last error on stdout; every rank also
# writes a short line to stderr, and all exit non-zero.  The first 8-GPU run of this file happens without anybody watching it.
STATE = {"stage": "start", "rank": 0, "world": 1, "rccl_ranks": 0, "eng": None, "reported": False, "args": None}
STAGE_TIMEOUT_S = float(os.environ.get("GENOMAD_AMD_BENCH_STAGE_TIMEOUT", "1500"))


def set_stage(name: str):
    """cpu_baseline / engine / comm_init / warmup / timed / gather / parity / encoder / main_e2e / metagenome / done"""
    import signal
    STATE["stage"] = name
    if hasattr(signal, "alarm"):
        signal.alarm(0 if name == "done" else int(STAGE_TIMEOUT_S))
    if os.environ.get("GENOMAD_AMD_BENCH_TEST_FAIL") == f"{STATE['rank']}:{name}":      # tests only: this rank dies on entering the stage
        raise RuntimeError(f"GENOMAD_AMD_BENCH_TEST_FAIL: rank {STATE['rank']} dies on entering stage {name}")


def _error_marker():
    from genomad_amd import rccl
    return rccl._rdzv_dir() / f"bench_error_{rccl._run_tag(0).hex()}"


def report_failure(kind: str, detail: str, code: int = 4):
    if STATE["reported"]:
        os._exit(code)
    STATE["reported"] = True
    last = ""
    try:
        if STATE["eng"] is not None:
            e = STATE["eng"].lib.gnn_last_error()
            last = e.decode(errors="replace") if isinstance(e, bytes) else str(e or "")
    except Exception:  # noqa: BLE001
        pass
    rec = {"error": kind, "rank": STATE["rank"], "world": STATE["world"], "stage": STATE["stage"], "detail": detail[-2000:],
           "rccl_ranks": STATE["rccl_ranks"], "gnn_last_error": last, "metric": "6 kbp windows classified/sec", "value": None,
           "n_gpus": STATE["world"], "hint": {"comm_init": "a rank never joined: see its stderr; GENOMAD_AMD_RDZV_TIMEOUT is the wait",
                                              "gather": "RCCL collective: NCCL_DEBUG=WARN output of rank 0 is on stderr"}.get(STATE["stage"])}
    print(f"bench.py: rank {STATE['rank']} FAILED in stage {STATE['stage']}: {kind}: {detail[-500:]}", file=sys.stderr, flush=True)
    first = True
    try:
        fd = os.open(_error_marker(), os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        os.write(fd, json.dumps(rec).encode())
        os.close(fd)
    except FileExistsError:
        first = False
    except Exception:  # noqa: BLE001
        pass
    if first:
        print(json.dumps(rec), flush=True)
    os._exit(code)          # not sys.exit: a rank stuck inside a collective must not run destructors that wait for its peers


def install_failure_handlers():
    import signal
    import traceback

    def on_signal(signum, _frame):
        name = signal.Signals(signum).name
        if signum == getattr(signal, "SIGALRM", None):
            report_failure("stage_timeout", f"stage '{STATE['stage']}' ran longer than {STAGE_TIMEOUT_S:.0f} s (GENOMAD_AMD_BENCH_STAGE_TIMEOUT)", 5)
        report_failure("terminated", f"{name} from the launcher (another rank failed, or the job was cancelled)", 128 + signum)

    for sig in ("SIGTERM", "SIGINT", "SIGALRM", "SIGHUP"):
        if hasattr(signal, sig):
            signal.signal(getattr(signal, sig), on_signal)

    def on_exception(tp, val, tb):
        if tp is SystemExit:
            raise val
        report_failure(tp.__name__, "".join(traceback.format_exception(tp, val, tb)))
    sys.excepthook = on_exception


def write_fasta(path, seq, offsets, name=b"contig_%d"):
    """`seq` (uint8) cut at `offsets` into records with 80-column lines, written with a few large writes per record."""
    import numpy as np
    with open(path, "wb") as f:
        for i in range(len(offsets) - 1):
            body = seq[int(offsets[i]):int(offsets[i + 1])]
            f.write(b">" + (name % i if b"%d" in name else name) + b" len=%d\n" % len(body))
            full = len(body) // 80 * 80
            if full:
                lines = np.empty((full // 80, 81), np.uint8)
                lines[:, :80] = body[:full].reshape(-1, 80)
                lines[:, 80] = 10
                f.write(lines.tobytes())
            if full < len(body):
                f.write(body[full:].tobytes() + b"\n")


def main_e2e_block(eng, weights, big_mb: int = 600):
    """Untimed block of the N = 1 line: genomad_amd.nn_classification.main() with the reference's signature
    (input_path, output_path, single_window, batch_size, restart, threads, verbose, cleanup: nn_classification.py:21-30) on
    (a) a config-1-shaped genome (one 5.1 Mbp chromosome + the seven plasmids of the documented lengths; seeded
        synthetic content): seconds, windows/s, and the TSV it wrote against the oracle chain (reference windowing rules -> fp32 forward ->
        segment mean -> the reference's row format, nn_classification.py:344-348) for the seven plasmids (56 windows; the CPU oracle
        needs ~25 ms per window, the chromosome's 850 windows are checked against the exact-f32 device path instead);
    (b) a ~600 MB synthetic metagenome FASTA (the synthetic window stream cut into 1-500 kbp contigs) written to the temp directory:
        seconds of the second (warm) call, windows/s, MB/s of FASTA text, the parity sentinel's value from the log."""
    import re
    import shutil
    import tempfile
    from pathlib import Path
    import numpy as np
    from genomad_amd import nn_classification as nnc, sequence, synthetic, weights as W
    from oracle import igloo_oracle, sequence_oracle          # the checker of (a)'s TSV, nothing timed
    tmp = Path(tempfile.mkdtemp(prefix="genomad_amd_bench_e2e_"))
    prev_engine, prev_w = nnc._ENGINE, os.environ.get("GENOMAD_AMD_WEIGHTS")
    out = {"entry_point": "genomad_amd.nn_classification.main(input_path, output_path, single_window=False, batch_size=128, restart=True, "
                          "threads=1, verbose=False, cleanup=False)  [reference signature: modules/nn_classification.py:21-30]"}
    try:
        wpath = tmp / "weights.npz"
        W.save_npz(wpath, weights)
        os.environ["GENOMAD_AMD_WEIGHTS"] = str(wpath)
        nnc._ENGINE = eng                                     # the bench's engine (same weights): no second 1.4 GB table, no second context

        def run(fa, tag):
            t = time.perf_counter()
            nnc.main(fa, tmp / tag, False, 128, True, 1, False, False)
            dt = time.perf_counter() - t
            d = tmp / tag / f"{fa.stem}_nn_classification"
            z = np.load(d / f"{fa.stem}_nn_classification.npz")
            wid = np.load(d / f"{fa.stem}_encoded_sequences" / f"{fa.stem}_seq_window_id.npz")
            log = (tmp / tag / f"{fa.stem}_nn_classification.log").read_text()
            m = re.search(r"Parity sentinel: max \|dscore\| of (\S+) .*? = ([0-9.eE+-]+)", log)
            return dt, z, len(wid["contig_ids"]), d, (float(m.group(2)) if m else None), (m.group(1) if m else None)

        # (a) config 1
        lengths = [5_100_000, 82_240, 61_331, 51_887, 50_635, 44_850, 28_729, 5_251]
        rng = np.random.default_rng(1895)
        fa = tmp / "GCF_009025895.1.fna"
        seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), sum(lengths))
        with open(fa, "wb") as f:
            pos = 0
            for i, n in enumerate(lengths):
                write_fasta_record = seq[pos:pos + n]
                f.write(b">NZ_CP0450%d.1 Klebsiella pneumoniae (synthetic stand-in)\n" % (15 + i))
                body = write_fasta_record.tobytes()
                f.write(b"\n".join(body[j:j + 80] for j in range(0, n, 80)) + b"\n")
                pos += n
        dt, z, nwin, d, sent, arith = run(fa, "config1")
        tsv = (d / f"{fa.stem}_nn_classification.tsv").read_bytes().decode().splitlines()
        plasmids = tmp / "plasmids.fna"
        text = fa.read_bytes()
        plasmids.write_bytes(text[text.index(b">NZ_CP045016.1"):])
        names, ids, wins = sequence_oracle.encode_fasta(plasmids)
        want = sequence_oracle.segment_mean(igloo_oracle.classify_windows(wins, weights, np.float32), ids)
        want_rows = [f"{n}\t" + "\t".join(f"{x:.4f}" for x in row) for n, row in zip(names, want)]          # nn_classification.py:344-348
        got_rows = tsv[2:]
        printed = np.array([[float(x) for x in r.split("\t")[1:]] for r in got_rows])
        _, cseq, coff = sequence.read_fasta_packed(fa)
        exact, _ = eng.classify_contigs(cseq, coff, False, "f32")
        out["config1"] = {
            "fasta": "8 records shaped like GCF_009025895.1 (5.1 Mbp chromosome + 7 plasmids), seeded synthetic ACGT, 80-column lines",
            "fasta_mb": round(fa.stat().st_size / 1e6, 2), "contigs": int(len(z["contig_names"])), "windows": int(nwin),
            "seconds": round(dt, 3), "windows_per_s": round(nwin / dt, 1), "parity_sentinel": sent, "arithmetic": arith,
            "tsv_header_ok": tsv[0] == "seq_name\tchromosome_score\tplasmid_score\tvirus_score" and len(tsv) == 9,
            "tsv_names_ok": [r.split("\t")[0] for r in tsv[1:]] == [f"NZ_CP0450{15 + i}.1" for i in range(8)],
            "tsv_plasmid_rows_byte_equal_to_oracle": int(sum(a == b for a, b in zip(got_rows, want_rows))), "tsv_plasmid_rows": len(want_rows),
            "tsv_max_abs_diff_vs_oracle": float(np.abs(printed - want).max()),
            "max_abs_dscore_vs_oracle_plasmids": float(np.abs(z["predictions"][1:] - want).max()),
            "max_abs_dscore_vs_exact_f32_all_contigs": float(np.abs(z["predictions"] - exact).max()),
            "oracle": "reference windowing rules (sequence_oracle.encode_fasta) -> numpy fp32 forward -> segment mean, rows formatted '%.4f' "
                      "tab-separated (modules/nn_classification.py:344-348); 56 windows"}
        c1 = out["config1"]
        # a printed value is the score rounded to 4 decimals: two scores within the 1e-4 tolerance print at most one unit (1e-4) apart
        c1["tsv_matches_oracle"] = bool(c1["tsv_header_ok"] and c1["tsv_names_ok"] and c1["tsv_max_abs_diff_vs_oracle"] <= 1.0001e-4
                                        and c1["max_abs_dscore_vs_oracle_plasmids"] <= 1e-4 and c1["max_abs_dscore_vs_exact_f32_all_contigs"] <= 1e-4)
        # (b) a FASTA of about big_mb MB: bases synthesised on the device and brought back (the CPU generator needs ~15 s for this size)
        nwin_big = big_mb * 1_000_000 // 6000
        dev = eng.alloc(nwin_big * 6000)
        try:
            eng.synth_windows_dev(0, nwin_big, dev.ptr)
            eng.sync()
            stream = dev.download((nwin_big * 6000,), np.uint8)
        finally:
            dev.free()
        off = synthetic.synth_metagenome_offsets(len(stream), seed=5)
        big = tmp / "meta.fna"
        t = time.perf_counter()
        write_fasta(big, stream, off)
        t_write = time.perf_counter() - t
        size = big.stat().st_size
        runs = []
        for k in range(2):
            dt, z, nw, _, sent, arith = run(big, f"big{k}")
            runs.append(dt)
        out["metagenome_fasta"] = {
            "fasta": f"{len(off) - 1} contigs of 1-500 kbp cut from the synthetic window stream (N runs and N tails included), 80-column lines, "
                     f"written to the temp directory in {t_write:.1f} s",
            "fasta_mb": round(size / 1e6, 1), "contigs": int(len(z["contig_names"])), "windows": int(nw),
            "seconds_first_call": round(runs[0], 3), "seconds": round(runs[1], 3), "windows_per_s": round(nw / runs[1], 1),
            "fasta_mb_per_s": round(size / 1e6 / runs[1], 1), "parity_sentinel": sent, "arithmetic": arith,
            "note": "whole calls of main(): FASTA validation, md5 of the input for the execution-info JSON (one sequential pass, ~1 GB/s), native "
                    "packing, H2D overlapped with the classification, per-contig means, NPZ / TSV / JSON written; the second call is the figure "
                    "(the first grows the library's staging buffers)"}
    finally:
        nnc._ENGINE = prev_engine
        if prev_w is None:
            os.environ.pop("GENOMAD_AMD_WEIGHTS", None)
        else:
            os.environ["GENOMAD_AMD_WEIGHTS"] = prev_w
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def metagenome_block(eng, precision: str, gbp: float = 3.0, gbp_per_chunk: float = 0.6, check_windows: int = 8000):
    """Untimed block of the N = 1 line: `gbp` Gbp of BASELINE configs[4] (mixed 1-500 kbp contigs, streamed through HBM in chunks
    synthesised on the device, spans -> N rule -> encode + IGLOO -> per-contig mean on the device) - windows/s - and one 48 Mbp
    chunk against the same rules applied on the host + the window path, bit for bit (what tests/test_gpu_parity.py::test_config5_...
    asserts)."""
    import numpy as np
    from genomad_amd import sequence, synthetic
    chunk_bytes = int(gbp_per_chunk * 1e9) // 6000 * 6000
    n_chunks = max(1, int(round(gbp * 1e9 / chunk_bytes)))
    seq = eng.alloc(chunk_bytes)
    try:
        def run_chunk(c):
            eng.synth_windows_dev(c * (chunk_bytes // 6000), chunk_bytes // 6000, seq.ptr)
            offsets = synthetic.synth_metagenome_offsets(chunk_bytes, seed=synthetic.DATA_SEED + c)
            pr, ids = eng.classify_contigs_dev(seq.ptr, offsets, False, precision)
            return offsets, pr, ids
        run_chunk(0)                                   # warm: the contig workspace grows once
        eng.sync()
        t0 = time.perf_counter()
        n_windows = n_contigs = n_bp = 0
        for c in range(n_chunks):
            offsets, pr, ids = run_chunk(c)
            n_windows, n_contigs, n_bp = n_windows + len(ids), n_contigs + len(pr), n_bp + int(offsets[-1])
        eng.sync()
        dt = time.perf_counter() - t0
        # one reduced chunk, bit for bit against host rules + window path
        nwin = check_windows
        offsets = synthetic.synth_metagenome_offsets(nwin * 6000, seed=99)
        eng.synth_windows_dev(0, nwin, seq.ptr)
        eng.sync()
        got, ids = eng.classify_contigs_dev(seq.ptr, offsets, False, precision)
        host = seq.download((nwin * 6000,), np.uint8)
        starts, lens, cids, wn = sequence.candidate_spans(offsets)
        keep = np.array([wn[i] == 0 or np.count_nonzero(host[starts[i]:starts[i] + lens[i]] == ord("N")) <= 4000 for i in range(len(starts))])
        wins = np.full((int(keep.sum()), 6000), ord("N"), np.uint8)
        for r, i in enumerate(np.flatnonzero(keep)):
            wins[r, :lens[i]] = host[starts[i]:starts[i] + lens[i]]
        want = eng.segment_mean(eng.classify(wins, precision), cids[keep], len(offsets) - 1)
        return {"gbp": round(n_bp / 1e9, 3), "contigs": int(n_contigs), "windows": int(n_windows), "seconds": round(dt, 3),
                "windows_per_s": round(n_windows / dt, 1), "gbp_per_s": round(n_bp / 1e9 / dt, 3), "chunks": n_chunks,
                "workload": f"BASELINE.json configs[4] at {gbp:g} Gbp on one GPU: contigs of 1-500 kbp (log-uniform), chunks of {chunk_bytes / 1e9:.2f} Gbp "
                            "synthesised in HBM, spans -> N rule -> encode+IGLOO -> per-contig mean on the device (`--workload metagenome --gbp-total 60` runs the full size)",
                "bit_equal_to_window_path": bool(np.array_equal(ids, cids[keep]) and np.array_equal(got, want)),
                "bit_equal_check": f"{nwin * 6000 / 1e6:.0f} Mbp chunk ({len(offsets) - 1} contigs, {int(keep.sum())} windows kept of {len(keep)}): device front end == "
                                   "host restatement of nn_classification.py:66-73 + gnn_classify + gnn_segment_mean"}
    finally:
        seq.free()


def spawn_ranks(n: int, cmd, env=None) -> int:
    """Run `cmd` as n local ranks (one per GPU) the way a launcher would: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in the environment, plus a private rendezvous directory and nonce for the RCCL unique id
    (genomad_amd/rccl.py).  Rank 0 inherits stdout (its one JSON line is the output); the other ranks' stdout goes to
    stderr.  Returns the first non-zero exit code (the remaining ranks are terminated by pid), else 0."""
    import secrets
    import shutil
    import socket
    import subprocess
    import tempfile
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    rdzv = tempfile.mkdtemp(prefix="genomad_amd_bench_")          # 0700, ours
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GENOMAD_AMD_RDZV_DIR=rdzv,
                GENOMAD_AMD_RDZV_NONCE=secrets.token_hex(8), GENOMAD_AMD_RDZV_PARENT=str(os.getpid()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    try:
        for r in range(n):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen(list(cmd), env=e, stdout=None if r == 0 else sys.stderr))
        rc, live = 0, set(range(n))
        while live and rc == 0:
            for r in sorted(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0:
                        rc = code
                        print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                        break
            time.sleep(0.05)
        return rc
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.terminate()
        for p_ in procs:
            try:
                p_.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p_.kill()
        # the failing rank's ONE error line (report_failure): rank 0 owns this process's stdout and printed its own; another rank's
        # went to stderr with the rest of its output, so the launcher repeats it on stdout from the marker file
        import glob
        for m in glob.glob(os.path.join(rdzv, "bench_error_*")):
            try:
                rec = json.loads(open(m).read())
                if rec.get("rank") != 0:
                    print(json.dumps(rec), flush=True)
            except Exception:  # noqa: BLE001
                pass
        shutil.rmtree(rdzv, ignore_errors=True)


def make_engine(local_rank: int, weights, chunk: int):
    """(engine, visible device count, local_rank used).  GENOMAD_AMD_BENCH_FAKE_ENGINE=1 swaps in tests/fake_engine.py — a CPU
    stand-in that exists so that the multi-rank plumbing of this file (spawn, shards, gather, the line's fields) can be
    exercised on a box without GPUs; its line says so ("fake_engine": true) and is not a measurement."""
    if os.environ.get("GENOMAD_AMD_BENCH_FAKE_ENGINE") == "1":
        from tests import fake_engine
        return fake_engine.FakeEngine(local_rank, weights, chunk), 8, local_rank
    from genomad_amd import _lib
    from genomad_amd.engine import NNEngine
    n_dev = ctypes.c_int()
    _lib.check(_lib.load().gnn_device_count(ctypes.byref(n_dev)))
    if local_rank >= max(n_dev.value, 1):
        # fewer visible GPUs than ranks (e.g. `--gpus 2` on a one-GPU box): the rank shares device r mod D, so that the launch
        # runs as far as RCCL's own duplicate-device check instead of stopping at the device index
        print(f"bench.py: local rank {local_rank}: {n_dev.value} device(s) visible, using device {local_rank % max(n_dev.value, 1)} "
              f"(RCCL refuses two ranks on one device)", file=sys.stderr)
        local_rank %= max(n_dev.value, 1)
    return NNEngine(local_rank, weights, chunk=chunk), n_dev.value, local_rank


def main():
    from genomad_amd._lib import DEFAULT_PRECISION
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows-per-step", type=int, default=65536, help="windows per step of the whole job (strong) or of every rank (weak): "
                    "16 x 65536 = the 1 M windows of BASELINE configs[2]/[3]; at 8 ranks a step is still 8192 windows = 32 rounds of workgroups "
                    "per rank (one GPU: 187.8-188.0 k windows/s for 16384, 32768 and 65536 per step, profiles/r04/backend_overlap_ab.txt)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--chunk", type=int, default=16384, help="windows per launch of the fused kernel (183.9 k windows/s against 180.3 k with "
                    "launches of 4096 on one box - the launch's tail is amortised over 64 instead of 16 rounds of workgroups per CU)")
    ap.add_argument("--precision", default="auto", choices=["auto", "f16c6", "f16x3", "f16x3tc", "f16x3tk", "bf16x3", "f32"],
                    help=f"arithmetic of the fused front end (default auto: f16x3tk when every rank's device can hold the k-mer tables - 156 GB "
                         f"built in 1.5 .. 6 s, outside the timed region - else {DEFAULT_PRECISION}; both have margin inside the 1e-4 tolerance; "
                         "f16c6 is faster than f16x3tc and exceeds it on a few of 10^6 windows)")
    ap.add_argument("--no-kmer-tables", action="store_true", help="--precision auto: do not build the k-mer tables (measure f16x3tc)")
    ap.add_argument("--async-steps", action="store_true",
                    help="step with gnn_classify_dev_async (the last back end of a step runs beside the next step's front end) "
                         "instead of the synchronous gnn_classify_dev that main() uses.  Measured equal for the default arithmetic "
                         "(141.4 vs 141.1 k windows/s at 2048 windows per step: the kernel is power-bound, the overlapped back end "
                         "costs the front end what it saves), +8-10 %% for the opt-in f16c6 at that step size")
    ap.add_argument("--check", default="all", choices=["all", "golden", "none"],
                    help="untimed parity pass: 'all' = every timed window against the exact-f32 device path (+ the golden "
                         "file), 'golden' = only the committed reference-graph scores of the first 10 000 windows")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="windows for the CPU baseline (0 = skip)")
    ap.add_argument("--kernel", default="classify", choices=["classify", "encoder"],
                    help="'encoder' benches the stand-alone byte->one-hot HBM kernel instead")
    ap.add_argument("--onehot-dtype", default="u8", choices=["u8", "bf16", "f32"])
    ap.add_argument("--workload", default="windows", choices=["windows", "metagenome"],
                    help="'metagenome' = BASELINE configs[4]: mixed 1-500 kbp contigs -> window spans -> N rule -> "
                         "encode+IGLOO -> per-contig mean, contigs sharded over the ranks")
    ap.add_argument("--gbp-total", type=float, default=None, help="metagenome: Gbp of the whole job (default: "
                    "steps x gbp-per-step per GPU); BASELINE configs[4] is 60")
    ap.add_argument("--gbp-per-step", type=float, default=0.6, help="metagenome: Gbp resident in HBM per step and GPU")
    ap.add_argument("--force-dist", action="store_true", help="(kept for old command lines: the communicator is always created now)")
    ap.add_argument("--fast-mode-steps", type=int, default=0,
                    help="after the timed region, time this many steps with the opt-in fast arithmetic f16c6 as well and report it "
                         "beside the default's line (default 0 = skip: the mode cannot ship - it leaves the 1e-4 tolerance on 10^6 windows); it "
                         "never enters `value`")
    ap.add_argument("--power", action="store_true",
                    help="sample `rocm-smi --showpower` of this rank's GPU on a helper thread during the timed region: mean watts and "
                         "joules per window in the line")
    ap.add_argument("--no-coalesce", action="store_true",
                    help="strong scaling with several ranks: launch every step of a rank on its own (65536 / N windows: 8192 at N = 8 = 32 rounds "
                         "of workgroups) instead of coalescing a rank's consecutive steps into launches of --chunk windows (the default: the "
                         "job's K steps are still all classified inside the timed region, there is still ONE gather; VERDICT r04 item 7)")
    ap.add_argument("--dump-scores", default=None, metavar="PATH.npy",
                    help="rank 0 saves the gathered (total, 3) f32 scores of the timed steps (tests/test_multi_gpu.py compares a D-rank run "
                         "with a 1-rank run of the same job bit for bit)")
    ap.add_argument("--no-encoder", action="store_true", help="skip the encoder block of the classification line (untimed, < 0.2 s)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed main_e2e and metagenome blocks of the N = 1 line (~25 s)")
    ap.add_argument("--share-devices", action="store_true",
                    help="testing aid for boxes with fewer GPUs than ranks: rank r uses device r mod (visible devices), so the "
                         "spawn and the RCCL bootstrap run as far as RCCL's own duplicate-device check")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # a plain process asked for N GPUs: become the launcher of N ranks of this same command line
        sys.exit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    import numpy as np
    from genomad_amd import _lib, rccl, sharding, synthetic

    rccl.prepare_env()
    rank, world, local_rank = rccl.world_from_env()
    args.gpus = world                    # under a launcher the launcher's world is the truth
    STATE.update(rank=rank, world=world, args=args)
    install_failure_handlers()
    # RCCL's own warnings (a peer that cannot be reached, a duplicate device) go to stderr, never into the one JSON line on stdout
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
    set_stage("cpu_baseline")

    weights = synthetic.synth_weights()
    # The CPU baseline runs on rank 0 at EVERY N (north_star: "next to the reference CPU path timed on the same box's host cores in
    # the same run"), before the engine and the communicator exist: the other ranks wait for rank 0's RCCL unique id meanwhile
    # (RcclComm polls for up to 10 minutes), no GPU work is in flight and nothing timed has started.
    cpu_base, cpu_scores = None, None
    if rank == 0 and args.cpu_sample > 0 and args.kernel == "classify":
        cpu_base, cpu_scores = cpu_baseline(weights, args.cpu_sample)
        if args.workload == "metagenome":
            cpu_base["sample"] += (" The metagenome's windows are cut from this same synthetic byte stream (read flat), so the sample is "
                                   "a sample of its windows; the CPU side of contig cutting / the N rule / the per-contig mean is negligible next to it.")
    set_stage("engine")
    eng, _n_dev, local_rank = make_engine(local_rank, weights, args.chunk)
    STATE["eng"] = eng
    info = eng.device_info()
    set_stage("comm_init")
    t_ci = time.perf_counter()
    comm = rccl.RcclComm(eng, rank, world, timeout=float(os.environ.get('GENOMAD_AMD_RDZV_TIMEOUT', '600')))   # always: the N = 1 line takes the same RCCL path as N = 8
    comm_init_s = time.perf_counter() - t_ci
    rccl_ranks, rccl_rank = ctypes.c_int(), ctypes.c_int()
    _lib.check(eng.lib.gnn_comm_info(eng.ctx, ctypes.byref(rccl_ranks), ctypes.byref(rccl_rank)))
    STATE["rccl_ranks"] = int(rccl_ranks.value)
    assert rccl_ranks.value == world and rccl_rank.value == rank
    # which physical GPU every rank is bound to (PCI bus id): two ranks on one device are visible in the line
    my_dev = (eng.pci_bus_id() if hasattr(eng, "pci_bus_id") else f"fake:{local_rank}").encode()
    gathered_dev_ids = sharding.gather_bytes(comm, my_dev)                 # a collective: every rank calls it, rank 0 gets the list
    per_rank_device = [b.decode() for b in gathered_dev_ids] if gathered_dev_ids is not None else None

    def barrier():
        eng.sync()
        comm.barrier()
        eng.sync()

    # ---- the arithmetic of this run (outside every timed region): f16x3tk needs its tables on EVERY rank
    kmer = None
    if args.precision in ("auto", "f16x3tk") and args.kernel == "classify":
        set_stage("kmer_tables")
        t_k = time.perf_counter()
        built = False
        if not (args.precision == "auto" and args.no_kmer_tables) and hasattr(eng, "build_kmer_tables"):
            built = bool(eng.build_kmer_tables())
        all_built = comm.allreduce_max(0.0 if built else 1.0) == 0.0
        if args.precision == "f16x3tk" and not all_built:
            raise RuntimeError("--precision f16x3tk: a rank's device cannot hold the k-mer tables (156 GB + workspaces)")
        if built and not all_built:
            eng.drop_kmer_tables()
        kmer = {"built": all_built, "seconds": round(time.perf_counter() - t_k, 2),
                "gb": round(eng.lib.gnn_kmer_tables_bytes() / 1e9, 1) if all_built else 0.0,
                "note": "x2 per 14-mer (137.4 GB), head A's pair products per (entry, 9-mer) (8.8 GB), conv2's tap tables for the rows no "
                        "14-mer indexes (8.3 GB), x1 over head A's index space (1.4 GB); built by gnn_build_kmer_tables before anything is timed; "
                        "most of the seconds are hipMalloc"}
        args.precision = "f16x3tk" if all_built else DEFAULT_PRECISION

    def max_over_ranks(x: float) -> float:
        return comm.allreduce_max(x)

    wps, K = args.windows_per_step, args.steps

    if args.kernel == "encoder":
        # stand-alone encoder: bases -> one-hot (n,5997,257); HBM-write bound
        wps_enc = min(wps, 2048)
        isz = {"u8": 1, "bf16": 2, "f32": 4}[args.onehot_dtype]
        code = {"u8": _lib.OH_U8, "bf16": _lib.OH_BF16, "f32": _lib.OH_F32}[args.onehot_dtype]
        bases = eng.alloc(wps_enc * 6000)
        out = eng.alloc(wps_enc * 5997 * 257 * isz)
        eng.synth_windows_dev(rank * wps_enc, wps_enc, bases.ptr)
        for _ in range(max(args.warmup, 1)):
            _lib.check(eng.lib.gnn_onehot_dev(eng.ctx, bases.ptr, wps_enc, code, out.ptr))
        eng.profile_enable(True)
        eng.profile_reset()
        barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            _lib.check(eng.lib.gnn_onehot_dev(eng.ctx, bases.ptr, wps_enc, code, out.ptr))
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        ms, launches = eng.profile_get(_lib.K_ENCODER)
        if rank == 0:
            per = ENCODER_BYTES[args.onehot_dtype] * wps_enc
            gbs = per * launches / (ms * 1e-3) / 1e9
            print(json.dumps({
                "metric": "6 kbp windows one-hot encoded/sec (stand-alone encoder kernel)",
                "value": round(world * wps_enc * K / dt, 1), "unit": "windows/s", "n_gpus": world, "steps": K,
                "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": args.onehot_dtype, "data": "synthetic",
                "config": {"workload": f"{wps_enc} synthetic 6 kbp windows per step -> one-hot depth 257 ({args.onehot_dtype})"},
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                             "kernel": "onehot_kernel", "bytes_per_launch": per,
                             "avg_launch_ms": round(ms / max(launches, 1), 5)}}))
        return

    if args.workload == "metagenome":
        # BASELINE configs[4]: the contigs of the job are dealt to the ranks in contiguous runs (a contig
        # never straddles ranks: per-contig means are local).  Every rank streams its run through HBM in
        # chunks of --gbp-per-step: the chunk's bytes are synthesised on the device (the synthetic-window
        # byte stream read flat, so it contains N runs and N-padded tails), cut into contigs of log-uniform
        # length (seeded per chunk), and go through the whole contig front end: window spans (numpy index
        # math), the N-content rule, upper-case/pad + encode + IGLOO, per-contig mean on the device.  Chunk
        # k+1 is synthesised while chunk k is classified (same stream: it is part of the job's input
        # feed).  The per-contig scores of all ranks are gathered on rank 0 at the end.
        chunk_bytes = int(args.gbp_per_step * 1e9) // 6000 * 6000
        total_bytes = int(args.gbp_total * 1e9) if args.gbp_total else chunk_bytes * K * world
        n_chunks_all = max(1, -(-total_bytes // chunk_bytes))
        my_chunks = [c for c in range(n_chunks_all) if c * world // n_chunks_all == rank] if n_chunks_all >= world \
            else ([rank] if rank < n_chunks_all else [])
        seq = eng.alloc(chunk_bytes)

        def run_chunk(c):
            eng.synth_windows_dev(c * (chunk_bytes // 6000), chunk_bytes // 6000, seq.ptr)
            offsets = synthetic.synth_metagenome_offsets(chunk_bytes, seed=synthetic.DATA_SEED + c)
            pr, ids = eng.classify_contigs_dev(seq.ptr, offsets, False, args.precision)
            return offsets, pr, ids

        for _ in range(min(args.warmup, 1)):
            run_chunk(my_chunks[0] if my_chunks else 0)
        eng.profile_enable(True)
        eng.profile_reset()
        barrier()
        t0 = time.perf_counter()
        parts, n_windows, n_bp = [], 0, 0
        for c in my_chunks:
            offsets, pr, ids = run_chunk(c)
            parts.append((c, np.zeros(len(pr), dtype="<U1"), pr, ids))
            n_windows += len(ids)
            n_bp += int(offsets[-1])
        _, preds, _, total_windows = sharding.gather_contig_parts(comm, parts)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        tot = comm.allgather_i64([n_bp]).sum()
        front_ms, front_launches = eng.profile_get(_lib.K_F32_FRONT if args.precision == "f32" else _lib.K_FUSED)
        back_ms, _ = eng.profile_get(_lib.K_BACKEND)
        eng.profile_enable(False)
        if rank == 0:
            extra = {}
            if front_launches and args.precision != "f32":
                extra["roofline"] = front_roofline(args.precision, n_windows / front_launches, front_ms / front_launches, front_launches,
                                                   front_ms, back_ms)
                extra["roofline"]["launch_note"] = (f"rank 0: {n_windows} windows in {front_launches} launches of the fused front end (a chunk's windows "
                                                    f"in launches of at most {args.chunk}); achieved = mean over those launches")
            if cpu_base is not None:
                extra["cpu_baseline"] = cpu_base
            print(json.dumps({
                "metric": "6 kbp windows classified/sec", "value": round(total_windows / dt, 1),
                "unit": "windows/s", "n_gpus": world, "steps": len(my_chunks), "warmup": min(args.warmup, 1),
                "ms_per_step": round(dt / max(len(my_chunks), 1) * 1e3, 2), "higher_is_better": True,
                "scaling": "strong" if args.gbp_total else "weak",
                "vs_baseline": None, "dtype": DTYPE_TEXT[args.precision], "data": "synthetic",
                "gbp_per_s": round(float(tot) / dt / 1e9, 3), "seconds": round(dt, 3),
                "config": {"workload": f"synthetic metagenome of {float(tot) / 1e9:.2f} Gbp, {len(preds)} contigs of 1-500 kbp "
                                       f"(log-uniform) = {total_windows} windows, contigs sharded over {world} GPU(s), "
                                       f"streamed through HBM in chunks of {chunk_bytes / 1e9:.2f} Gbp synthesised on the "
                                       f"device; spans -> N rule -> encode+IGLOO -> per-contig mean on the device, "
                                       f"per-contig scores gathered on rank 0 (BASELINE.json configs[4]"
                                       f"{'' if args.gbp_total and args.gbp_total >= 60 else ' at reduced size'}); "
                                       f"5 kernels per {args.chunk}-window launch, launch overhead < 0.1 %, no hipGraph",
                           "precision": args.precision, "contigs": int(len(preds)),
                           "mean_contig_score": [round(float(x), 6) for x in preds.mean(axis=0)]}, **extra}))
        comm.close()
        return

    # ---- windows workload (BASELINE configs[2]/[3])
    if args.scaling == "strong":
        if wps % world:
            sys.exit(f"--windows-per-step {wps} must be divisible by the number of ranks {world}")
        wps_local = wps // world
        total = wps * K
        first = rank * (total // world)          # contiguous shard [first, first + n_local) of the job's windows
    else:
        wps_local = wps
        total = wps * K * world
        first = rank * wps * K
    n_local = wps_local * K
    bases = eng.alloc(max(n_local * 6000, 1))
    scores = eng.alloc(max(n_local * 12, 1))
    gathered_dev = eng.alloc(max(total * 12, 1)) if rank == 0 else None
    for k in range(K):
        eng.synth_windows_dev(first + k * wps_local, wps_local, bases.ptr + k * wps_local * 6000)
    eng.sync()

    classify_step = eng.classify_dev_async if args.async_steps else eng.classify_dev
    # Launch shape.  A rank's steps are contiguous in its shard, so with several ranks in strong mode (a step =
    # 65536 / N windows per rank: 8192 at N = 8, half the tuned launch of 16384) consecutive steps are handed to the library together,
    # `group` at a time, and it cuts them into launches of --chunk windows as it does with a 65536-window step at N = 1.  All K steps
    # of the job are classified inside the timed region either way; --no-coalesce launches every step on its own.
    group = 1 if (args.no_coalesce or wps_local >= args.chunk or wps_local == 0) else max(1, args.chunk // wps_local)

    def step(k, g=1):
        classify_step(bases.ptr + k * wps_local * 6000, g * wps_local, scores.ptr + k * wps_local * 12, args.precision)

    def all_steps():
        for k in range(0, K, group):
            step(k, min(group, K - k))

    set_stage("warmup")
    for i in range(0, args.warmup, group):
        step((i % K) if (i % K) + group <= K else 0, min(group, K))
    eng.profile_enable(True)
    eng.profile_reset()
    sampler = PowerSampler(local_rank) if args.power else None
    barrier()
    set_stage("timed")
    t0 = time.perf_counter()
    all_steps()
    eng.flush()
    eng.sync()
    t_own = time.perf_counter() - t0             # this rank's own K steps (reported per rank; `value` uses the max below)
    watts = sampler.stop() if sampler is not None else []
    set_stage("gather")
    # ONE gather of every rank's (n_local, 3) f32 scores to rank 0 (ncclGather over xGMI; at N = 1 the same call) ...
    t_g = time.perf_counter()
    comm.gather_dev(scores.ptr, gathered_dev.ptr if gathered_dev is not None else None, n_local * 12, 0)
    host_scores = None
    if rank == 0:                 # ... and on to the host of rank 0, still inside the timed region
        host_scores = gathered_dev.download((total, 3), np.float32)
    else:
        eng.sync()
    gather_ms = (time.perf_counter() - t_g) * 1e3     # on rank 0: includes waiting for the slowest rank's last step
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    own_us = comm.allgather_i64([int(t_own * 1e6)])[:, 0]
    # the same gather once more, untimed and between barriers (no rank is still computing): what the collective itself costs,
    # so that a slow first RCCL call cannot be mistaken for poor scaling
    barrier()
    t_g2 = time.perf_counter()
    comm.gather_dev(scores.ptr, gathered_dev.ptr if gathered_dev is not None else None, n_local * 12, 0)
    eng.sync()
    gather_ms_isolated = max_over_ranks((time.perf_counter() - t_g2) * 1e3)

    kid = _lib.K_F32_FRONT if args.precision == "f32" else _lib.K_FUSED
    front_ms, front_launches = eng.profile_get(kid)
    back_ms, _ = eng.profile_get(_lib.K_BACKEND)
    eng.profile_enable(False)
    front_us_ranks = comm.allgather_i64([int(front_ms * 1e3), int(front_launches)])

    # ---- untimed: the opt-in fast arithmetic on the same steps, for the record (f16c6: 1.5 MFMA pass equivalents; it does NOT hold
    # the 1e-4 tolerance on 10^6 windows, which is why it is not the default and never enters `value`)
    mine = scores.download((n_local, 3), np.float32)
    fast = None
    if args.fast_mode_steps > 0 and args.precision != "f16c6":
        kf = min(args.fast_mode_steps, K)
        for k in range(min(2, kf)):
            eng.classify_dev(bases.ptr + k * wps_local * 6000, wps_local, scores.ptr + k * wps_local * 12, "f16c6")
        eng.profile_enable(True)
        eng.profile_reset()
        barrier()
        tf = time.perf_counter()
        for k in range(kf):
            classify_step(bases.ptr + k * wps_local * 6000, wps_local, scores.ptr + k * wps_local * 12, "f16c6")
        eng.flush()
        barrier()
        dtf = max_over_ranks(time.perf_counter() - tf)
        fms, fl = eng.profile_get(_lib.K_FUSED)
        eng.profile_enable(False)
        fast_scores = scores.download((kf * wps_local, 3), np.float32)
        fast = {"precision": "f16c6", "value": round(kf * wps_local * world / dtf, 1), "unit": "windows/s", "steps": kf,
                "avg_launch_ms": round(fms / max(fl, 1), 4),
                "frac": round(FLOP_PER_WINDOW * (kf * wps_local / max(fl, 1)) / (fms / max(fl, 1) * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4) if fms > 0 else None,
                "max_abs_dscore_vs_default_this_rank0": float(np.abs(fast_scores - mine[:kf * wps_local]).max()),
                "note": "opt-in (--precision f16c6 / GENOMAD_AMD_PRECISION=f16c6): 1.5 MFMA pass equivalents; 8e-5 on the 10 000-window parity "
                        "config but 1.2e-4 on a few of 10^6 windows - no head-room under the 1e-4 tolerance, so not the default"}

    set_stage("parity")
    # ---- untimed checks of what was just timed, on every rank over ITS windows
    # (1) bit for bit against a second run through the synchronous entry point
    for k in range(K):
        eng.classify_dev(bases.ptr + k * wps_local * 6000, wps_local, scores.ptr + k * wps_local * 12, args.precision)
    eng.sync()
    again = scores.download((n_local, 3), np.float32)
    mismatching = int(comm.allgather_i64([int(np.count_nonzero((mine != again).any(axis=1)))])[:, 0].sum())
    # (2) every timed window against the exact-f32 device path (itself 4e-6 from the fp64 oracle and 1.1e-5 from the
    #     reference-graph golden on config 2: tests/test_gpu_parity.py): max |dscore|, and how many windows pass 1e-4 / 5e-5
    parity = None
    if args.check == "all" and args.precision != "f32":
        t_chk = time.perf_counter()
        for k in range(K):
            eng.classify_dev(bases.ptr + k * wps_local * 6000, wps_local, scores.ptr + k * wps_local * 12, "f32")
        eng.sync()
        exact = scores.download((n_local, 3), np.float32)
        d = np.abs(mine - exact).max(axis=1) if n_local else np.zeros(0, np.float32)
        finite = bool(np.isfinite(mine).all())
        worst = float(d.max()) if len(d) else 0.0
        cnt = comm.allgather_i64([int((d > 1e-4).sum()), int((d > 5e-5).sum()), int(not finite), int(n_local)]).sum(axis=0)
        parity = {"max_abs_dscore_all": max_over_ranks(worst if np.isfinite(worst) else 1e9), "windows": int(cnt[3]),
                  "over_1e-4": int(cnt[0]), "over_5e-5": int(cnt[1]), "non_finite_ranks": int(cnt[2]), "tolerance": 1e-4,
                  "against": "exact-f32 device path (GNN_PREC_F32) on every timed window of every rank, untimed",
                  "seconds": round(time.perf_counter() - t_chk, 1)}
        parity["ok"] = parity["over_1e-4"] == 0 and parity["non_finite_ranks"] == 0 and parity["max_abs_dscore_all"] <= 1e-4
    failed = []
    if mismatching:
        failed.append(f"{mismatching} windows differ bit-wise between the timed steps and their synchronous re-run")
    if parity is not None and not parity["ok"]:
        failed.append(f"{parity['over_1e-4']} of {parity['windows']} timed windows exceed the 1e-4 score tolerance against the "
                      f"exact-f32 path (max {parity['max_abs_dscore_all']:.3e})")

    if rank == 0:
        out = {
            "metric": "6 kbp windows classified/sec", "value": round(total / dt, 1), "unit": "windows/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": DTYPE_TEXT[args.precision], "data": "synthetic",
            "rccl_ranks": int(rccl_ranks.value),
            "per_rank_windows_per_s": [round(n_local / (u * 1e-6), 1) for u in own_us.tolist()],
            "per_rank_steps_ms": [round(u * 1e-3, 2) for u in own_us.tolist()],
            "per_rank_front_ms_total": [round(u * 1e-3, 2) for u in front_us_ranks[:, 0].tolist()],
            "gather_ms": round(gather_ms, 3), "gather_ms_isolated": round(gather_ms_isolated, 3),
            "comm_init_s": round(comm_init_s, 3), "per_rank_device": per_rank_device,
            "config": {"workload": f"{total} synthetic 6 kbp windows ({K} steps x {wps}{' per GPU' if args.scaling == 'weak' else ''}), "
                                   f"{'sharded contiguously over' if args.scaling == 'strong' else 'on each of'} {world} GPU(s) "
                                   f"= {n_local} per GPU, synthetic weights of the reference shapes, HBM-resident input, "
                                   f"scores gathered to rank 0 with one ncclGather over {int(rccl_ranks.value)} RCCL rank(s) and "
                                   f"copied to its host (BASELINE.json configs[2]/[3]); 5 kernels per "
                                   f"{min(args.chunk, group * wps_local)}-window launch, launch overhead < 0.1 %, no hipGraph"
                                   + (f"; a rank's consecutive steps are handed to the library {group} at a time (one launch shape at every N)" if group > 1 else ""),
                       "precision": args.precision, "windows_per_launch": min(args.chunk, group * wps_local), "steps_per_call": group,
                       "entry_point": "gnn_classify_dev_async" if args.async_steps else "gnn_classify_dev",
                       "device": info["name"].strip(), "cus": info["cus"]},
        }
        win_per_launch = n_local / max(front_launches, 1)
        avg_ms = front_ms / max(front_launches, 1)
        tflops = FLOP_PER_WINDOW * win_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        passes = MFMA_PASSES.get(args.precision)
        out["roofline"] = front_roofline(args.precision, win_per_launch, avg_ms, front_launches, front_ms, back_ms)
        if args.precision != "f32":
            # context for `frac`: what this power-managed chip sustains on the bf16 MFMA with nothing else running
            # (outside the timed region, ~200 ms)
            probe = ctypes.c_double()
            _lib.check(eng.lib.gnn_mfma_probe(eng.ctx, 200, ctypes.byref(probe)))
            out["roofline"]["issued_mfma_tflops_bf16_equivalent"] = round(tflops * passes, 1)
            out["roofline"]["mfma_probe_sustained_tflops"] = round(probe.value, 1)
            out["roofline"]["issued_vs_probe"] = round(tflops * passes / probe.value, 4)
            if args.precision == "f16c6":
                # the power floor of the opt-in fast arithmetic: its own MFMA mix (8 f16 + 4 MX-fp6 per k32 step) on register operands,
                # in algorithmic TFLOP/s
                same = ctypes.c_double()
                _lib.check(eng.lib.gnn_mfma_probe_kind(eng.ctx, 2, 200, ctypes.byref(same)))
                out["roofline"]["mfma_probe_same_mix_algorithmic_tflops"] = round(same.value, 1)
                out["roofline"]["frac_of_power_floor"] = round(tflops / same.value, 4)
                out["roofline"]["frac_ceiling_at_power_floor"] = round(same.value / MFMA_PEAK_TFLOPS, 4)
            if args.precision in ("f16x3", "f16x3tc", "f16x3tk", "bf16x3"):
                # the power floor of THIS arithmetic: register-operand MFMAs of the kernel's own instruction (f16 draws more than
                # bf16 per MFMA on this chip), nothing else running; the kernel issues `passes` of them per algorithmic product
                same = ctypes.c_double()
                _lib.check(eng.lib.gnn_mfma_probe_kind(eng.ctx, 0 if args.precision == "bf16x3" else 1, 200, ctypes.byref(same)))
                out["roofline"]["mfma_probe_same_instruction_tflops"] = round(same.value, 1)
                out["roofline"]["frac_of_power_floor"] = round(tflops * passes / same.value, 4)
                out["roofline"]["frac_ceiling_at_power_floor"] = round(same.value / passes / MFMA_PEAK_TFLOPS, 4)
        # parity of what was just timed: every window against the exact-f32 device path (above), and the first windows of
        # the job against the committed outputs of the reference's own graph (tests/golden/config2_golden.npz, windows 0..9999)
        real = os.environ.get("GENOMAD_AMD_BENCH_FAKE_ENGINE") != "1"
        if not args.no_encoder and real:
            set_stage("encoder")
            try:
                out["encoder"] = encoder_block(eng, rank)
            except Exception as exc:  # noqa: BLE001   an untimed extra must not cost the measured line
                out["encoder"] = {"skipped": f"{type(exc).__name__}: {exc}"}
        # the drop-in entry point and config 5 under the driver's clock: N = 1 only, untimed, after everything
        # that enters `value`; a failure is reported inside its block and in `failed`
        if kmer is not None:
            out["kmer_tables"] = kmer
        if world == 1 and real and not args.no_extras and args.precision in (DEFAULT_PRECISION, "f16x3tk"):
            for name, fn in (("main_e2e", lambda: main_e2e_block(eng, weights)), ("metagenome", lambda: metagenome_block(eng, args.precision))):
                set_stage(name)
                t_x = time.perf_counter()
                try:
                    out[name] = fn()
                    out[name]["block_seconds"] = round(time.perf_counter() - t_x, 1)
                except Exception as exc:  # noqa: BLE001
                    import traceback
                    out[name] = {"error": f"{type(exc).__name__}: {exc}", "traceback": traceback.format_exc()[-1500:]}
                    failed.append(f"{name} block: {type(exc).__name__}: {exc}")
            if "config1" in out.get("main_e2e", {}) and not out["main_e2e"]["config1"]["tsv_matches_oracle"]:
                failed.append("main_e2e: the TSV main() wrote does not match the oracle chain within 1e-4")
            if "bit_equal_to_window_path" in out.get("metagenome", {}) and not out["metagenome"]["bit_equal_to_window_path"]:
                failed.append("metagenome: the contig front end is not bit-equal to host rules + window path")
        if fast is not None:
            out["fast_mode"] = fast
        if watts:
            w_mean = sum(watts) / len(watts)
            out["power"] = {"mean_watts_rank0": round(w_mean, 1), "samples": len(watts), "joules_per_window": round(w_mean * world * dt / total, 5),
                            "source": "rocm-smi --showpower of rank 0's GPU sampled during the timed region (x world for the job)"}
        if parity is not None:
            out["parity"] = parity
        gpath = os.path.join(ROOT, "tests", "golden", "config2_golden.npz")
        if os.path.exists(gpath) and args.check != "none":
            g = np.load(gpath)["scores_refgraph32"]
            m = min(len(g), len(host_scores) if args.scaling == "strong" or world == 1 else n_local)
            out["max_abs_dscore"] = float(np.abs(host_scores[:m] - g[:m]).max())
            out["dscore_windows"] = int(m)
            out["dscore_reference"] = "tests/golden/config2_golden.npz: reference create_classifier() graph, float32"
            out["dscore_tolerance"] = 1e-4
            if not out["max_abs_dscore"] <= 1e-4:
                failed.append(f"max |dscore| {out['max_abs_dscore']:.3e} against the reference-graph golden exceeds 1e-4")
        out["steps_verified"] = {"windows_all_ranks": int(total), "mismatching_windows_all_ranks": mismatching,
                                 "against": "untimed re-run of every step through the synchronous gnn_classify_dev, bit for bit"}
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
            if args.scaling == "strong" or world == 1:      # host_scores[:sample] are the job's first windows = the CPU sample
                out["max_abs_dscore_vs_cpu_baseline"] = float(np.abs(host_scores[:len(cpu_scores)] - cpu_scores).max())
        if args.dump_scores:
            np.save(args.dump_scores, host_scores)
        if os.environ.get("GENOMAD_AMD_BENCH_FAKE_ENGINE") == "1":
            out["fake_engine"] = True
            out["data"] = "FAKE ENGINE on the CPU (tests/fake_engine.py): a test of this file's multi-rank plumbing, not a measurement"
        if failed:
            out["failed"] = failed
        print(json.dumps(out), flush=True)
    set_stage("done")
    comm.close()
    if failed:
        if rank == 0:
            print("bench.py: FAILED: " + "; ".join(failed), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
