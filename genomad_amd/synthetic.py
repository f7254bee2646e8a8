"""Seeded synthetic inputs: 6 kbp windows and weights of the exact reference shapes.

The trained weights (genomad/data/nn_classifier.h5) are absent from the reference
checkout, so benchmarks and parity tests run on seeded synthetic weights with the
shapes/dtypes of the reference model (genomad/neural_network/model.py:14-45,
igloo.py:117-188).  The window generator is counter based so the same windows can
be produced on the host (numpy, here) and on the device (csrc/gnn_synth.hip).
"""
import numpy as np

WINDOW = 6000
N_TOKENS = 5997
N_PATCHES = 2100
PATCH_SIZE = 4
N_CH = 128
POOL = 8
N_POOLED = N_TOKENS // POOL  # 749, igloo.py:176
DATA_SEED = 1234
WEIGHT_SEED = 42

_M64 = (1 << 64) - 1


def splitmix64(x):
    """splitmix64 finaliser on uint64 numpy arrays (wrap-around arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def window_thresholds(iv, seed):
    """Per-window base-composition thresholds (tA, tC, tG) on a 16-bit uniform.

    r = splitmix64(seed + i + 2**25); GC content gc/65536 in [0.25, 0.75); A takes
    sA/65536 in [0.375, 0.625) of the AT share, C takes sC/65536 of the GC share.
    Integer arithmetic only, so host and device agree bit for bit.
    """
    with np.errstate(over="ignore"):
        r = splitmix64(np.uint64(seed) + iv + np.uint64(1 << 25))
    gc = np.uint64(16384) + (r & np.uint64(0x7FFF))
    sA = np.uint64(24576) + ((r >> np.uint64(16)) & np.uint64(0x3FFF))
    sC = np.uint64(24576) + ((r >> np.uint64(32)) & np.uint64(0x3FFF))
    tA = ((np.uint64(65536) - gc) * sA) >> np.uint64(16)
    tC = tA + ((gc * sC) >> np.uint64(16))
    tG = tA + gc
    return tA, tC, tG


def synth_windows(first: int, count: int, seed: int = DATA_SEED) -> np.ndarray:
    """Windows ``first .. first+count`` of the synthetic data set, (count, 6000) uint8 ASCII.

    u(i, p) = splitmix64(seed ^ (i*6000 + p)) >> 48          (16-bit uniform)
    base    = A if u < tA else C if u < tC else G if u < tG else T, thresholds per window
              (window_thresholds) so that windows differ in composition and scores vary
    i % 16 == 5 : true length L = 2500 + splitmix64(seed + i) % 3501, positions >= L are 'N'
                  (the reference's right padding, nn_classification.py:72)
    i % 64 == 9 : one internal run of 'N': r = splitmix64(seed + i + 2**24),
                  len = 1 + r % 200, start = (r >> 16) % (6000 - len)
    """
    i = np.arange(first, first + count, dtype=np.uint64)[:, None]
    p = np.arange(WINDOW, dtype=np.uint64)[None, :]
    u = splitmix64(np.uint64(seed) ^ (i * np.uint64(WINDOW) + p)) >> np.uint64(48)
    iv = i[:, 0]
    tA, tC, tG = window_thresholds(iv, seed)
    idx = (u >= tA[:, None]).astype(np.int64) + (u >= tC[:, None]) + (u >= tG[:, None])
    out = np.frombuffer(b"ACGT", dtype=np.uint8)[idx]
    with np.errstate(over="ignore"):
        L = np.uint64(2500) + splitmix64(np.uint64(seed) + iv) % np.uint64(3501)
        r = splitmix64(np.uint64(seed) + iv + np.uint64(1 << 24))
    run_len = np.uint64(1) + r % np.uint64(200)
    run_off = (r >> np.uint64(16)) % (np.uint64(WINDOW) - run_len)
    pp = p.astype(np.int64)
    short = ((iv % np.uint64(16)) == 5)[:, None] & (pp >= L.astype(np.int64)[:, None])
    run = ((iv % np.uint64(64)) == 9)[:, None] & (pp >= run_off.astype(np.int64)[:, None]) \
        & (pp < (run_off + run_len).astype(np.int64)[:, None])
    out = np.where(short | run, np.uint8(78), out)
    return np.ascontiguousarray(out, dtype=np.uint8)


def gen_patches(rng: np.random.Generator, n_patches=N_PATCHES, patch_size=PATCH_SIZE,
                vector_size=N_TOKENS) -> np.ndarray:
    """Patch indices with the distribution of igloo.py:280-301 (return_sequences=False,
    build_backbone=False): each patch = ``patch_size`` distinct positions drawn uniformly
    from [0, vector_size), sorted ascending.  Shape (n_patches, patch_size, 1) int32."""
    out = np.empty((n_patches, patch_size, 1), dtype=np.int32)
    for k in range(n_patches):
        out[k, :, 0] = np.sort(rng.choice(vector_size, size=patch_size, replace=False))
    return out


def _glorot(rng, shape, fan_in, fan_out, gain=1.0):
    lim = gain * np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def synth_weights(seed: int = WEIGHT_SEED) -> dict:
    """Synthetic weights, keys = the repo's weight schema (see genomad_amd/weights.py).

    Scales are chosen so that parity tests are non-vacuous: attention is peaked
    (max/mean of alpha >> 1), class scores vary across windows, logits stay bounded.
    """
    rng = np.random.default_rng(seed)
    w = {}
    w["conv1_kernel"] = _glorot(rng, (6, 257, N_CH), 6, N_CH, gain=2.0)   # one-hot input: 6 active rows
    w["conv1_bias"] = rng.uniform(-0.1, 0.1, N_CH).astype(np.float32)
    for name in ("conv2", "conv3"):
        w[f"{name}_kernel"] = _glorot(rng, (6, N_CH, N_CH), 6 * N_CH, N_CH, gain=2.0)
        w[f"{name}_bias"] = rng.uniform(-0.1, 0.1, N_CH).astype(np.float32)
    for head in ("iglooA", "iglooB"):
        w[f"{head}_patches"] = gen_patches(rng)
        w[f"{head}_w_mult"] = rng.normal(0, 1.0, (1, N_PATCHES, PATCH_SIZE, N_CH)).astype(np.float32)
        w[f"{head}_w_summer"] = rng.normal(0, 1.0 / np.sqrt(PATCH_SIZE * N_CH),
                                           (1, PATCH_SIZE * N_CH, 1)).astype(np.float32)
        w[f"{head}_w_bias"] = rng.normal(0, 0.5, (1, N_PATCHES)).astype(np.float32)
        w[f"{head}_w_qk"] = rng.normal(0, 2.0 / np.sqrt(N_PATCHES), (N_PATCHES, N_POOLED)).astype(np.float32)
        w[f"{head}_w_v"] = _glorot(rng, (1, N_CH, N_CH), N_CH, N_CH)
    for name, (fi, fo) in (("enc", (2 * N_CH, 512)), ("head", (512, 512))):
        w[f"{name}_dense_kernel"] = _glorot(rng, (fi, fo), fi, fo, gain=1.5)
        w[f"{name}_dense_bias"] = rng.uniform(-0.1, 0.1, fo).astype(np.float32)
        w[f"{name}_bn_gamma"] = rng.uniform(0.5, 1.5, fo).astype(np.float32)
        w[f"{name}_bn_beta"] = rng.normal(0, 0.2, fo).astype(np.float32)
        w[f"{name}_bn_mean"] = rng.normal(0, 0.2, fo).astype(np.float32)
        w[f"{name}_bn_var"] = rng.uniform(0.5, 2.0, fo).astype(np.float32)
    w["out_dense_kernel"] = _glorot(rng, (512, 3), 512, 3, gain=3.0)
    # constants that roughly centre the three logits over the synthetic windows, so that
    # the class scores spread over (0, 1) instead of saturating on one class
    w["out_dense_bias"] = np.array([-8.95, 6.15, -2.82], dtype=np.float32)
    return w


def synth_metagenome_offsets(total_bp: int, seed: int = DATA_SEED, min_len: int = 1_000, max_len: int = 500_000):
    """Contig layout of BASELINE config 5: contig lengths log-uniform in
    [1 kbp, 500 kbp] (seeded), packed back to back until ``total_bp`` bytes are used; the last
    contig is cut to fit (and dropped if that leaves it shorter than ``min_len``).  Returns the
    (n_contigs+1,) int64 offsets.  The bytes themselves are the synthetic-window stream
    (:func:`synth_windows` / ``gnn_synth_windows_dev``) read as one flat buffer."""
    rng = np.random.default_rng(seed ^ 0x5EED)
    est = max(16, int(total_bp / 60_000))
    lens = []
    used = 0
    while used < total_bp:
        batch = np.exp(rng.uniform(np.log(min_len), np.log(max_len), est)).astype(np.int64)
        for L in batch:
            L = int(min(L, total_bp - used))
            if L < min_len:
                used = total_bp
                break
            lens.append(L)
            used += L
            if used >= total_bp:
                break
    return np.concatenate([[0], np.cumsum(np.asarray(lens, dtype=np.int64))]).astype(np.int64)
