"""Weight schema of the nn classifier and its marshalling into the C ABI.

Schema = flat ``.npz`` whose arrays keep the reference's Keras layouts
(genomad/neural_network/model.py:14-45, igloo.py:117-188):

    conv1_kernel (6,257,128) conv1_bias (128,)            igloo.py:45-47 on the one-hot input
    conv2_kernel conv3_kernel (6,128,128) + *_bias        igloo.py:66, loop iterations 1 and 2
    iglooA_* / iglooB_*: patches (2100,4,1) int32, w_mult (1,2100,4,128), w_summer (1,512,1),
        w_bias (1,2100), w_qk (2100,749), w_v (1,128,128) igloo.py:129-188 (A on conv1, B on conv3)
    enc_dense_kernel (256,512) enc_dense_bias, enc_bn_{gamma,beta,mean,var} (512,)   model.py:28-29
    head_dense_kernel (512,512) ...                                                  model.py:40-41
    out_dense_kernel (512,3) out_dense_bias (3,)                                     model.py:44

The trained reference blob (genomad/data/nn_classifier.h5, Keras legacy H5) is not in
the reference checkout; ``load_npz`` reads this repo's schema.
"""
import ctypes as C

import numpy as np

from . import _lib

SHAPES = {
    "conv1_kernel": (6, 257, 128), "conv1_bias": (128,),
    "conv2_kernel": (6, 128, 128), "conv2_bias": (128,),
    "conv3_kernel": (6, 128, 128), "conv3_bias": (128,),
    "enc_dense_kernel": (256, 512), "enc_dense_bias": (512,),
    "head_dense_kernel": (512, 512), "head_dense_bias": (512,),
    "out_dense_kernel": (512, 3), "out_dense_bias": (3,),
}
for _h in ("iglooA", "iglooB"):
    SHAPES.update({f"{_h}_patches": (2100, 4, 1), f"{_h}_w_mult": (1, 2100, 4, 128),
                   f"{_h}_w_summer": (1, 512, 1), f"{_h}_w_bias": (1, 2100),
                   f"{_h}_w_qk": (2100, 749), f"{_h}_w_v": (1, 128, 128)})
for _l in ("enc", "head"):
    SHAPES.update({f"{_l}_bn_{s}": (512,) for s in ("gamma", "beta", "mean", "var")})


def validate(weights: dict) -> dict:
    """Return a dict of C-contiguous arrays of the schema's dtypes/shapes, or raise ValueError."""
    out = {}
    for name, shape in SHAPES.items():
        if name not in weights:
            raise ValueError(f"weights: missing tensor {name!r}")
        a = np.asarray(weights[name])
        if tuple(a.shape) != shape:
            raise ValueError(f"weights: {name} has shape {a.shape}, expected {shape}")
        if name.endswith("_patches"):
            # Keras stores the non-trainable index tensor as int32 (igloo.py:129-135); accept floats too
            if not np.all(a == np.round(a)):
                raise ValueError(f"weights: {name} is not integral")
            a = np.ascontiguousarray(a, dtype=np.int32)
            if a.min() < 0 or a.max() >= _lib.TOKENS:
                raise ValueError(f"weights: {name} index out of range [0, {_lib.TOKENS})")
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            if not np.all(np.isfinite(a)):
                raise ValueError(f"weights: {name} contains non-finite values")
        out[name] = a
    return out


def load_npz(path) -> dict:
    with np.load(path) as z:
        return validate({k: z[k] for k in z.files})


def save_npz(path, weights: dict) -> None:
    np.savez(path, **validate(weights))


def to_struct(w: dict):
    """(gnn_weights struct, keep-alive list).  ``w`` must come from :func:`validate`."""
    def fp(name):
        return w[name].ctypes.data_as(C.POINTER(C.c_float))

    def igloo(h):
        return _lib.IglooWeights(
            w[f"{h}_patches"].ctypes.data_as(C.POINTER(C.c_int32)), fp(f"{h}_w_mult"),
            fp(f"{h}_w_summer"), fp(f"{h}_w_bias"), fp(f"{h}_w_qk"), fp(f"{h}_w_v"))

    def dense(l):
        return _lib.DenseBN(fp(f"{l}_dense_kernel"), fp(f"{l}_dense_bias"), fp(f"{l}_bn_gamma"),
                            fp(f"{l}_bn_beta"), fp(f"{l}_bn_mean"), fp(f"{l}_bn_var"))

    s = _lib.Weights(fp("conv1_kernel"), fp("conv1_bias"), fp("conv2_kernel"), fp("conv2_bias"),
                     fp("conv3_kernel"), fp("conv3_bias"), igloo("iglooA"), igloo("iglooB"),
                     dense("enc"), dense("head"), fp("out_dense_kernel"), fp("out_dense_bias"))
    return s, w
