"""numpy restatement of the geNomad IGLOO classifier forward pass (fp32 or fp64).

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.  TF, Keras and the trained weights are
unavailable, so this is pinned against the reference's own model.py / igloo.py executed over numpy
stand-ins of the TF/Keras primitives (oracle/keras_shim.py; agreement 3e-15 in fp64) rather than
against a TensorFlow run.  The restatement follows the reference's op order and Keras defaults:

* Conv1D(128, 6, padding="causal"): stride 1, left zero pad of 5, cross-correlation,
  kernel (k, in, out), bias added                       igloo.py:45-47, :66
* LeakyReLU(negative_slope=0.1)                          igloo.py:48, :67
* SpatialDropout1D / Dropout: identity at inference      igloo.py:49-53, model.py:43
* IGLOO1D_kernel.call (transformer_style=True, incoming_proj=0)  igloo.py:190-214
* MaxPool1D(pool_size=8): stride 8, padding "valid"      igloo.py:209-210
* Dense: x @ kernel + bias; BatchNormalization(axis=-1, epsilon=1e-3) with moving
  statistics; relu; Dense(3, softmax)                    model.py:28-30, :40-44
* topology: conv1 -> IGLOO_A ; conv1 -> conv2 -> conv3 -> IGLOO_B ; concat
  (igloo.py:54-83: the second IGLOO1D_kernel call sits after the for loop)

Weights are a dict in the repo schema (genomad_amd/weights.py).
"""
import numpy as np

N_TOKENS = 5997
ONE_HOT_DEPTH = 257   # model.py:11
KSIZE = 6             # model.py:22
POOL = 8              # model.py:23
BN_EPS = 1e-3         # Keras BatchNormalization default; model.py:29,41 pass none
LRELU = 0.1           # igloo.py:48


def _lrelu(x):
    # LeakyReLU(0.1) = max(x, 0.1 x) for a slope below 1 (bit-identical to the select form, one pass less)
    return np.maximum(x, x * np.asarray(LRELU, dtype=x.dtype))


def _softmax(x):
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=-1, keepdims=True)


def conv1_gather(tokens, kernel, bias):
    """conv1 on a one-hot input == sum of 6 gathered kernel rows (model.py:11 + igloo.py:45-47).

    out[b,t,:] = bias + sum_{k, t+k-5>=0} kernel[k, tokens[b,t+k-5], :]
    """
    B, T = tokens.shape
    out = np.broadcast_to(bias, (B, T, kernel.shape[2])).copy()
    for k in range(KSIZE):
        shift = KSIZE - 1 - k          # tap k reads position t - shift
        if shift == 0:
            out += kernel[k][tokens]
        else:
            out[:, shift:, :] += kernel[k][tokens[:, :T - shift]]
    return out


def conv1_dense_onehot(tokens, kernel, bias, conv=None):
    """Reference-faithful conv1: materialise tf.one_hot(depth=257) and contract (model.py:11)."""
    B, T = tokens.shape
    oh = np.zeros((B, T, ONE_HOT_DEPTH), dtype=kernel.dtype)
    oh.reshape(B * T, ONE_HOT_DEPTH)[np.arange(B * T), np.asarray(tokens).reshape(-1)] = 1
    return (conv or causal_conv)(oh, kernel, bias)


def causal_conv(x, kernel, bias):
    """Keras Conv1D(padding='causal'): out[t] = bias + sum_k x[t+k-(K-1)] @ kernel[k], x[<0]=0."""
    B, T, C = x.shape
    K, _, F = kernel.shape
    xp = np.concatenate([np.zeros((B, K - 1, C), dtype=x.dtype), x], axis=1)
    s0, s1, s2 = xp.strides
    cols = np.lib.stride_tricks.as_strided(xp, shape=(B, T, K, C), strides=(s0, s1, s1, s2))
    out = cols.reshape(B * T, K * C) @ kernel.reshape(K * C, F)
    return out.reshape(B, T, F) + bias


def causal_conv_shifted(x, kernel, bias):
    """Same convolution as :func:`causal_conv` without the im2col copy: ONE (B*T, C) @ (C, K*F) product, then
    tap k's (B, T, F) slice is added into the rows it reaches (tap k of output row t reads input row
    t + k - (K-1)).  Used by the timed CPU baseline (bench.py); differs from causal_conv only in the f32
    summation order."""
    B, T, C = x.shape
    K, _, F = kernel.shape
    wide = np.ascontiguousarray(kernel.transpose(1, 0, 2).reshape(C, K * F))
    y = (np.ascontiguousarray(x).reshape(B * T, C) @ wide).reshape(B, T, K, F)
    out = np.empty((B, T, F), dtype=x.dtype)
    out[...] = bias
    for k in range(K):
        shift = K - 1 - k
        out[:, shift:] += y[:, :T - shift, k]
    return out


def igloo_kernel_literal(y, patches, w_mult, w_summer, w_bias, w_qk, w_v, taps=None):
    """IGLOO1D_kernel.call op for op (igloo.py:190-214), numpy in place of tf."""
    M = np.transpose(y, (1, 2, 0))                      # (T, C, B)          igloo.py:192
    M = M[patches[:, :, 0]]                             # gather_nd -> (P, 4, C, B)  :193
    mpi = np.transpose(M, (3, 0, 1, 2))                 # (B, P, 4, C)       :194
    mpi = w_mult * mpi                                  #                    :195
    mpi = mpi.reshape(-1, patches.shape[0], patches.shape[1] * y.shape[2])   # :196-199
    mpi = mpi @ w_summer                                # (B, P, 1)          :204
    mpi = mpi[..., 0] + w_bias                          #                    :205-206
    y_proj = y @ w_v                                    # (B, T, C)          :208
    Tp = y.shape[1] // POOL
    y_proj = y_proj[:, :Tp * POOL].reshape(y.shape[0], Tp, POOL, y.shape[2]).max(axis=2)   # :209-210
    alpha = _softmax(mpi @ w_qk)                        #                    :211-212
    out = (alpha[:, None, :] @ y_proj)[:, 0, :]         #                    :213-214
    if taps is not None:
        taps.update(m=mpi, alpha=alpha, yp=y_proj)
    return out


def igloo_kernel_closed(y, patches, w_mult, w_summer, w_bias, w_qk, w_v, taps=None):
    """Closed form used to design the device kernels; equals the literal form."""
    P, J = patches.shape[:2]
    C = y.shape[2]
    weff = w_mult[0] * w_summer[0, :, 0].reshape(J, C)[None]       # (P, J, C)
    g = y[:, patches[:, :, 0], :]                                   # (B, P, J, C)
    m = np.einsum("bpjc,pjc->bp", g, weff) + w_bias
    alpha = _softmax(m @ w_qk)
    yp = y @ w_v[0]
    Tp = y.shape[1] // POOL
    yp = yp[:, :Tp * POOL].reshape(y.shape[0], Tp, POOL, C).max(axis=2)
    out = np.einsum("bq,bqc->bc", alpha, yp)
    if taps is not None:
        taps.update(m=m, alpha=alpha, yp=yp)
    return out


def _bn(x, gamma, beta, mean, var):
    return gamma * (x - mean) / np.sqrt(var + np.asarray(BN_EPS, dtype=x.dtype)) + beta


def forward(tokens, weights, dtype=np.float32, dense_onehot=False, literal=True, return_taps=False,
            shifted_conv=False):
    """Scores (B, 3) for ``tokens`` (B, 5997) ints in [0, 256].

    dtype        np.float32 ("reference-like") or np.float64 ("truth")
    dense_onehot build the explicit (B, 5997, 257) one-hot like the reference does
    literal      use the op-for-op IGLOO kernel (else the closed form)
    return_taps  also return a dict of intermediates
                 (x1, x2, x3, mA, alphaA, ypA, mB, alphaB, ypB, f, h1, h2, logits)
    shifted_conv convolutions as six shifted matrix products instead of im2col (faster; bench.py baseline)
    """
    tokens = np.asarray(tokens, dtype=np.int64)
    w = {k: (np.asarray(v).astype(dtype) if np.asarray(v).dtype.kind == "f" else np.asarray(v))
         for k, v in weights.items()}
    conv = causal_conv_shifted if shifted_conv else causal_conv
    if dense_onehot:
        x1 = _lrelu(conv1_dense_onehot(tokens, w["conv1_kernel"], w["conv1_bias"], conv))
    else:
        x1 = _lrelu(conv1_gather(tokens, w["conv1_kernel"], w["conv1_bias"]))
    x2 = _lrelu(conv(x1, w["conv2_kernel"], w["conv2_bias"]))
    x3 = _lrelu(conv(x2, w["conv3_kernel"], w["conv3_bias"]))
    ig = igloo_kernel_literal if literal else igloo_kernel_closed
    tA, tB = {}, {}
    fA = ig(x1, w["iglooA_patches"], w["iglooA_w_mult"], w["iglooA_w_summer"], w["iglooA_w_bias"],
            w["iglooA_w_qk"], w["iglooA_w_v"], tA)
    fB = ig(x3, w["iglooB_patches"], w["iglooB_w_mult"], w["iglooB_w_summer"], w["iglooB_w_bias"],
            w["iglooB_w_qk"], w["iglooB_w_v"], tB)
    f = np.concatenate([fA, fB], axis=-1)
    h1 = np.maximum(_bn(f @ w["enc_dense_kernel"] + w["enc_dense_bias"], w["enc_bn_gamma"],
                        w["enc_bn_beta"], w["enc_bn_mean"], w["enc_bn_var"]), 0)
    h2 = np.maximum(_bn(h1 @ w["head_dense_kernel"] + w["head_dense_bias"], w["head_bn_gamma"],
                        w["head_bn_beta"], w["head_bn_mean"], w["head_bn_var"]), 0)
    logits = h2 @ w["out_dense_kernel"] + w["out_dense_bias"]
    scores = _softmax(logits)
    if not return_taps:
        return scores
    taps = dict(x1=x1, x2=x2, x3=x3, f=f, h1=h1, h2=h2, logits=logits,
                mA=tA["m"], alphaA=tA["alpha"], ypA=tA["yp"],
                mB=tB["m"], alphaB=tB["alpha"], ypB=tB["yp"])
    return scores, taps


def classify_windows(bases, weights, dtype=np.float32, batch=16, **kw):
    """bases (n, 6000) uint8 ASCII -> scores (n, 3); tokenises with the closed-form oracle."""
    from .sequence_oracle import tokenize_closed_form
    out = []
    for a in range(0, len(bases), batch):
        tok = tokenize_closed_form(bases[a:a + batch])
        out.append(forward(tok, weights, dtype=dtype, **kw))
    return np.concatenate(out) if out else np.zeros((0, 3), dtype=dtype)
