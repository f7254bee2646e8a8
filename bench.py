#!/usr/bin/env python
"""bench.py — 6 kbp windows classified/sec on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): synthetic 6 kbp windows (seed 1234) + synthetic weights (seed 42) of the
reference shapes, BASELINE.json configs[2]/[3]: by default 64 steps x 16384 windows = 1 M windows
per GPU, resident in HBM before the timed region.  A "step" is one pass of the whole hot path
(tokenise -> conv1..3 -> IGLOO heads -> dense/softmax) over one batch of windows.  Weak scaling: every
rank classifies its own 1 M windows, no data-path collective; one RCCL gather of the scores to rank 0
at the end (inside the timed region).  `value` = windows of all ranks / max-over-ranks wall time.

Extra objects in the JSON line:
  roofline      dominant kernel (fused front end): algorithmic FLOP per launch / HIP-event duration
                measured live on the library's stream, against the dense bf16 MFMA peak (2.5 PFLOP/s)
  cpu_baseline  the numpy restatement of the reference (oracle/, reference-faithful mode: explicit
                one-hot, dense conv1) timed on the host cores on a bounded sample; N=1, rank 0 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_WINDOW = 2_762_901_136        # conv2+conv3, w_v x2, IGLOO small terms, head
MFMA_PEAK_TFLOPS = 2500.0              # MI355X_MICROARCH.md: dense bf16 MFMA peak
ENCODER_BYTES = {"u8": 6000 + 5997 * 257, "bf16": 6000 + 5997 * 257 * 2, "f32": 6000 + 5997 * 257 * 4}
HBM_PEAK_GBS = 8000.0


def cpu_baseline(weights, sample: int):
    """Time the CPU restatement (the checker) on `sample` windows; return (dict, scores)."""
    import numpy as np
    from genomad_amd import synthetic
    from oracle import igloo_oracle, sequence_oracle
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:  # noqa: BLE001
        blas_threads = len(os.sched_getaffinity(0))
    bases = synthetic.synth_windows(0, sample)
    tok = sequence_oracle.tokenize_closed_form(bases)
    igloo_oracle.forward(tok[:2], weights, np.float32, dense_onehot=True)      # warm up BLAS
    t = time.perf_counter()
    scores = []
    for a in range(0, sample, 16):
        scores.append(igloo_oracle.forward(tok[a:a + 16], weights, np.float32, dense_onehot=True))
    dt = time.perf_counter() - t
    t2 = time.perf_counter()
    igloo_oracle.forward(tok[:16], weights, np.float32, dense_onehot=False)
    dt_alg = (time.perf_counter() - t2) / min(16, sample)
    return ({"value": round(sample / dt, 2), "unit": "windows/s", "cores": int(blas_threads),
             "kind": "port",
             "sample": f"{sample} synthetic windows, numpy fp32 restatement in reference-faithful mode "
                       f"(explicit 5997x257 one-hot, dense conv1), batch 16, OpenBLAS threads={blas_threads}; "
                       f"algorithmic mode (conv1 as gather): {1.0 / dt_alg:.1f} windows/s",
             "host_cpus_visible": len(os.sched_getaffinity(0))},
            np.concatenate(scores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows-per-step", type=int, default=16384)
    ap.add_argument("--chunk", type=int, default=4096, help="windows per launch of the fused kernel")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "bf16", "f32"])
    ap.add_argument("--cpu-sample", type=int, default=48, help="windows for the CPU baseline (0 = skip)")
    ap.add_argument("--kernel", default="classify", choices=["classify", "encoder"],
                    help="'encoder' benches the stand-alone byte->one-hot HBM kernel instead")
    ap.add_argument("--onehot-dtype", default="u8", choices=["u8", "bf16", "f32"])
    ap.add_argument("--workload", default="windows", choices=["windows", "metagenome"],
                    help="'metagenome' = BASELINE config 5 at a chosen size: mixed 1-500 kbp contigs resident in HBM "
                         "-> window spans -> N rule -> encode+IGLOO -> per-contig mean (one step = --gbp-per-step)")
    ap.add_argument("--gbp-per-step", type=float, default=0.6, help="metagenome workload: Gbp of contigs per step and GPU")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the nccl (RCCL) process group and run the barrier/gather path even with one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs a torch.distributed.run launch with {args.gpus} ranks")
        args.gpus = world

    # torch is plumbing here: device selection, barrier, the final RCCL gather.  It is imported
    # before the HIP library so that both share one HIP runtime in the process.
    import numpy as np
    import torch
    import torch.distributed as dist
    from genomad_amd import _lib, sharding, synthetic
    from genomad_amd.engine import NNEngine

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    weights = synthetic.synth_weights()
    eng = NNEngine(local_rank, weights, chunk=args.chunk)
    info = eng.device_info()

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    wps, K = args.windows_per_step, args.steps
    n_local = wps * K

    if args.kernel == "encoder":
        # stand-alone encoder: bases -> one-hot (n,5997,257); HBM-write bound
        wps_enc = min(wps, 2048)
        isz = {"u8": 1, "bf16": 2, "f32": 4}[args.onehot_dtype]
        code = {"u8": _lib.OH_U8, "bf16": _lib.OH_BF16, "f32": _lib.OH_F32}[args.onehot_dtype]
        bases = torch.empty(wps_enc * 6000, dtype=torch.uint8, device=dev)
        out = torch.empty(wps_enc * 5997 * 257 * isz, dtype=torch.uint8, device=dev)
        eng.synth_windows_dev(rank * wps_enc, wps_enc, bases.data_ptr())
        for _ in range(max(args.warmup, 1)):
            _lib.check(eng.lib.gnn_onehot_dev(eng.ctx, bases.data_ptr(), wps_enc, code, out.data_ptr()))
        eng.profile_enable(True)
        eng.profile_reset()
        barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            _lib.check(eng.lib.gnn_onehot_dev(eng.ctx, bases.data_ptr(), wps_enc, code, out.data_ptr()))
        barrier()
        dt = time.perf_counter() - t0
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
        # BASELINE config 5, sized by --gbp-per-step: every rank holds its own packed contig buffer in
        # HBM (the synthetic-window byte stream read flat, so it contains N runs and N-padded tails),
        # cut into contigs of log-uniform length.  One step = the whole contig front end over it:
        # window spans (numpy index math), the N-content rule, upper-case/pad + encode + IGLOO, and
        # the per-contig mean on the device.
        nbytes = int(args.gbp_per_step * 1e9) // 6000 * 6000
        seq = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        eng.synth_windows_dev(rank * (nbytes // 6000), nbytes // 6000, seq.data_ptr())
        offsets = synthetic.synth_metagenome_offsets(nbytes, seed=synthetic.DATA_SEED + rank)
        eng.sync()
        n_windows = 0
        contig_scores = None
        for _ in range(args.warmup):
            contig_scores, ids = eng.classify_contigs_dev(seq.data_ptr(), offsets, False, args.precision)
        barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            contig_scores, ids = eng.classify_contigs_dev(seq.data_ptr(), offsets, False, args.precision)
            n_windows += len(ids)
        barrier()
        dt = time.perf_counter() - t0
        counts = torch.tensor([float(n_windows), float(offsets[-1]) * K, dt], dtype=torch.float64, device=dev)
        if use_dist:
            tot = counts.clone()
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            dist.all_reduce(counts, op=dist.ReduceOp.MAX)
            n_all, bp_all, dt = float(tot[0]), float(tot[1]), float(counts[2])
        else:
            n_all, bp_all = float(counts[0]), float(counts[1])
        if rank == 0:
            print(json.dumps({
                "metric": "6 kbp windows classified/sec (contig front end)", "value": round(n_all / dt, 1),
                "unit": "windows/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
                "ms_per_step": round(dt / K * 1e3, 2), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                "gbp_per_s": round(bp_all / dt / 1e9, 3),
                "config": {"workload": f"synthetic metagenome, {len(offsets) - 1} contigs of 1-500 kbp (log-uniform), "
                                       f"{offsets[-1] / 1e9:.2f} Gbp per step and GPU, HBM-resident; spans -> N rule -> "
                                       f"encode+IGLOO -> per-contig mean (BASELINE.json configs[4] at reduced size)",
                           "contigs_per_gpu": len(offsets) - 1, "windows_per_step_per_gpu": n_windows // max(K, 1),
                           "mean_contig_score": [round(float(x), 6) for x in contig_scores.mean(axis=0)]}}))
        if use_dist:
            dist.destroy_process_group()
        return

    bases = torch.empty(n_local * 6000, dtype=torch.uint8, device=dev)
    scores = torch.zeros((n_local, 3), dtype=torch.float32, device=dev)
    first = rank * n_local                      # weak scaling: every rank has its own windows
    for k in range(K):
        eng.synth_windows_dev(first + k * wps, wps, bases.data_ptr() + k * wps * 6000)
    eng.sync()

    def step(k):
        eng.classify_dev(bases.data_ptr() + k * wps * 6000, wps, scores.data_ptr() + k * wps * 12, args.precision)

    for i in range(args.warmup):
        step(i % K)
    eng.profile_enable(True)
    eng.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for k in range(K):
        step(k)
    eng.sync()
    gathered = sharding.gather_scores(scores, n_local * world) if use_dist else scores
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    kid = _lib.K_F32_FRONT if args.precision == "f32" else _lib.K_FUSED
    front_ms, front_launches = eng.profile_get(kid)
    back_ms, _ = eng.profile_get(_lib.K_BACKEND)

    if rank == 0:
        total = n_local * world
        out = {
            "metric": "6 kbp windows classified/sec", "value": round(total / dt, 1), "unit": "windows/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16x3": "bf16x3 (split-bf16 MFMA, 3 passes, f32 accumulate)", "bf16": "bf16",
                      "f32": "f32"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"{n_local} synthetic 6 kbp windows per GPU ({K} steps x {wps}), "
                                   f"synthetic weights of the reference shapes, HBM-resident input, "
                                   f"scores gathered to rank 0 (BASELINE.json configs[2]/[3])",
                       "precision": args.precision, "windows_per_launch": min(args.chunk, wps),
                       "device": info["name"].strip(), "cus": info["cus"]},
        }
        win_per_launch = n_local / max(front_launches, 1)
        avg_ms = front_ms / max(front_launches, 1)
        tflops = FLOP_PER_WINDOW * win_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            # measured in separate rocprofv3 --pmc passes of this command (profiles/README.md); per launch
            per_window = json.load(open(tpath)).get("bytes_per_window", {}).get(args.precision)
            if per_window is not None:
                traffic = int(per_window * win_per_launch)
        out["roofline"] = {
            "bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tflops / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
            "kernel": "fused_front_kernel" if kid == _lib.K_FUSED else "f32 front end (5 kernels)",
            "flop_per_launch": int(FLOP_PER_WINDOW * win_per_launch), "avg_launch_ms": round(avg_ms, 4),
            "launches": int(front_launches),
            "mfma_passes": 3 if args.precision == "bf16x3" else 1,
            "note": "achieved counts ALGORITHMIC flops (2.763 GFLOP/window); bf16x3 issues 3 MFMA passes "
                    "per product, so issued-MFMA utilisation is 3x frac",
            "backend_ms_total": round(back_ms, 2), "front_ms_total": round(front_ms, 2)}
        if args.precision != "f32":
            # context for `frac`: what this power-managed chip sustains on the same MFMA instruction
            # with nothing else running (outside the timed region, ~200 ms)
            import ctypes
            probe = ctypes.c_double()
            _lib.check(eng.lib.gnn_mfma_probe(eng.ctx, 200, ctypes.byref(probe)))
            passes = out["roofline"]["mfma_passes"]
            out["roofline"]["issued_mfma_tflops"] = round(tflops * passes, 1)
            out["roofline"]["mfma_probe_sustained_tflops"] = round(probe.value, 1)
            out["roofline"]["issued_vs_probe"] = round(tflops * passes / probe.value, 4)
        if world == 1 and args.cpu_sample > 0:
            base, cpu_scores = cpu_baseline(weights, args.cpu_sample)
            gpu_first = gathered[:args.cpu_sample].cpu().numpy()
            out["cpu_baseline"] = base
            out["max_abs_dscore"] = float(np.abs(gpu_first - cpu_scores).max())
            out["dscore_tolerance"] = 1e-4
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
