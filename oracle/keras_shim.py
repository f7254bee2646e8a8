"""Numpy stand-ins for the few TensorFlow / Keras primitives the reference's network code calls,
so that genomad/neural_network/model.py and igloo.py can be EXECUTED IN PLACE without TensorFlow.

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.

TensorFlow and Keras are not installed in this image (and cannot be), so the reference's
floating-point half cannot run as shipped.  What can be done is the same trick
``reference_harness.py`` plays with numba: register minimal ``tensorflow`` and ``keras`` modules
whose primitives are written in numpy, and import the reference's modules unmodified.  Then

* the op SEQUENCE — which tensor is transposed, gathered, multiplied, reshaped, pooled, softmaxed
  in what order, how the blocks are stacked and concatenated (igloo.py:28-83, :190-217,
  model.py:9-45) — is the reference's own Python, run where it lies;
* each PRIMITIVE (causal Conv1D, Dense, BatchNormalization inference, MaxPool1D, one_hot,
  gather_nd, matmul, softmax ...) is a one-to-five-line numpy function below, following the
  documented Keras/TF semantics cited at each.

Only what the reference calls is provided; anything else raises AttributeError, which is what we
want (a silent default would hide a behaviour difference).

The functional API (``Input`` -> layers -> ``Model``) is reproduced with a tiny deferred graph:
calling a layer on a symbolic tensor records a node, ``Model.predict`` evaluates it on real arrays.
Layers are built on first use from the real input shape (Keras builds from the symbolic shape; the
shapes are the same).  ``add_weight`` does not initialise anything: it asks the installed
*weight provider* for the tensor of that layer (n-th instance of its class) and name, so a test
can run the reference graph on the same weights as the oracle and the device.
"""
import sys
import types

import numpy as np

_STATE = {"provider": None, "counters": {}, "dtype": np.float32}


# ----------------------------------------------------------------------------- deferred graph
class Sym:
    """A symbolic tensor: either a graph input or the output of ``op`` applied to ``parents``."""

    def __init__(self, op=None, parents=(), nested=False):
        self.op, self.parents, self.nested = op, parents, nested


def _is_sym(x):
    return isinstance(x, Sym) or (isinstance(x, (list, tuple)) and any(isinstance(e, Sym) for e in x))


def _evaluate(node, env):
    if id(node) in env:
        return env[id(node)]
    if node.op is None:
        raise RuntimeError("graph input without a value")
    if node.nested:
        args = [_evaluate(p, env) for p in node.parents]
    else:
        args = _evaluate(node.parents[0], env)
    env[id(node)] = out = node.op(args)
    return out


# ----------------------------------------------------------------------------- keras.layers
class Layer:
    """keras.layers.Layer: ``__call__`` builds once from the input shape, then runs ``call``."""

    def __init__(self, **kwargs):
        cls = type(self).__name__
        self._index = _STATE["counters"].get(cls, 0)
        _STATE["counters"][cls] = self._index + 1
        self.built = False
        self.trainable = True

    def add_weight(self, shape=None, initializer=None, trainable=True, regularizer=None, name=None, dtype=None):
        if _STATE["provider"] is None:
            raise RuntimeError("keras_shim: no weight provider installed")
        w = np.asarray(_STATE["provider"](type(self).__name__, self._index, name))
        if shape is not None and tuple(int(s) for s in shape) != tuple(w.shape):
            raise ValueError(f"{type(self).__name__}[{self._index}].{name}: provider gave {w.shape}, layer wants {shape}")
        if np.issubdtype(w.dtype, np.floating):
            w = w.astype(_STATE["dtype"])
        return w

    def build(self, input_shape):
        self.built = True

    def _run(self, x):
        if not self.built:
            shape = [tuple(np.shape(e)) for e in x] if isinstance(x, (list, tuple)) else tuple(np.shape(x))
            self.build(shape)
            self.built = True
        return self.call(x)

    def __call__(self, x):
        if _is_sym(x):
            if isinstance(x, (list, tuple)):
                return Sym(self._run, tuple(x), nested=True)
            return Sym(self._run, (x,))
        return self._run(x)


def Input(shape=None, dtype=None):
    return Sym()


class Conv1D(Layer):
    """keras.layers.Conv1D(filters, kernel_size, padding="causal"), channels-last, stride 1:
    cross-correlation with kernel (k, in, out); "causal" left-pads k-1 zeros so that output[t]
    depends on input[t-k+1 .. t] (Keras docs, Conv1D ``padding``)."""

    def __init__(self, filters, kernel_size, padding="valid"):
        super().__init__()
        if padding != "causal":
            raise NotImplementedError(padding)
        self.filters, self.k = filters, kernel_size

    def build(self, input_shape):
        self.kernel = self.add_weight(shape=(self.k, input_shape[-1], self.filters), name="kernel")
        self.bias = self.add_weight(shape=(self.filters,), name="bias")

    def call(self, x):
        b, t, c = x.shape
        xp = np.concatenate([np.zeros((b, self.k - 1, c), x.dtype), x], axis=1)
        out = np.zeros((b, t, self.filters), np.result_type(x.dtype, self.kernel.dtype))
        for j in range(self.k):
            out += xp[:, j:j + t, :] @ self.kernel[j]
        return out + self.bias


class Dense(Layer):
    """keras.layers.Dense: activation(x @ kernel + bias)."""

    def __init__(self, units, activation=None):
        super().__init__()
        self.units, self.activation = units, activation

    def build(self, input_shape):
        self.kernel = self.add_weight(shape=(input_shape[-1], self.units), name="kernel")
        self.bias = self.add_weight(shape=(self.units,), name="bias")

    def call(self, x):
        y = x @ self.kernel + self.bias
        if self.activation is None:
            return y
        if self.activation == "softmax":
            return _softmax(y)
        raise NotImplementedError(self.activation)


class BatchNormalization(Layer):
    """Inference mode: gamma * (x - moving_mean) / sqrt(moving_variance + epsilon) + beta with the
    Keras default epsilon = 1e-3 (the reference passes no arguments, model.py:29,41)."""

    def build(self, input_shape):
        n = input_shape[-1]
        self.gamma = self.add_weight(shape=(n,), name="gamma")
        self.beta = self.add_weight(shape=(n,), name="beta")
        self.moving_mean = self.add_weight(shape=(n,), name="moving_mean")
        self.moving_variance = self.add_weight(shape=(n,), name="moving_variance")

    def call(self, x):
        return self.gamma * (x - self.moving_mean) / np.sqrt(self.moving_variance + x.dtype.type(1e-3)) + self.beta


class LeakyReLU(Layer):
    def __init__(self, negative_slope=0.3):
        super().__init__()
        self.slope = negative_slope

    def call(self, x):
        return np.where(x >= 0, x, x * x.dtype.type(self.slope))


class Activation(Layer):
    def __init__(self, name):
        super().__init__()
        if name != "relu":
            raise NotImplementedError(name)

    def call(self, x):
        return np.maximum(x, 0)


class _Identity(Layer):
    """Dropout / SpatialDropout1D at inference."""

    def __init__(self, rate=None):
        super().__init__()

    def call(self, x):
        return x


class Dropout(_Identity):
    pass


class SpatialDropout1D(_Identity):
    pass


class Concatenate(Layer):
    def call(self, xs):
        return np.concatenate(list(xs), axis=-1)


class MaxPool1D(Layer):
    """keras.layers.MaxPool1D(pool_size): strides = pool_size, padding "valid" -> floor(T/pool) windows."""

    def __init__(self, pool_size=2):
        super().__init__()
        self.pool = pool_size

    def call(self, x):
        b, t, c = x.shape
        q = t // self.pool
        return x[:, :q * self.pool].reshape(b, q, self.pool, c).max(axis=2)


class Model(Layer):
    """keras.Model(inputs, outputs): callable on arrays (predict) and, like any layer, on a symbolic tensor."""

    def __init__(self, inputs=None, outputs=None):
        super().__init__()
        self.inputs, self.outputs = inputs, outputs
        self.layers = []          # the reference only iterates it to freeze the encoder (model.py:35-36)
        self.built = True

    def call(self, x):
        return _evaluate(self.outputs, {id(self.inputs): np.asarray(x)})

    def predict(self, x, **_):
        return self.call(x)


# ----------------------------------------------------------------------------- tensorflow
def _softmax(x):
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def _one_hot(x, depth, axis=-1):
    """tf.one_hot: float32 rows, all-zero for indices outside [0, depth)."""
    if axis != -1:
        raise NotImplementedError
    x = np.asarray(x).astype(np.int64)
    out = np.zeros(x.shape + (depth,), _STATE["dtype"])
    ok = (x >= 0) & (x < depth)
    idx = np.nonzero(ok)
    out[idx + (x[ok],)] = 1
    return out


def _gather_nd(params, indices):
    """tf.gather_nd with index depth 1: out[..., :] = params[indices[..., 0]]."""
    indices = np.asarray(indices)
    if indices.shape[-1] != 1:
        raise NotImplementedError
    return params[indices[..., 0]]


def _make_modules():
    tf = types.ModuleType("tensorflow")
    tf.matmul = lambda a, b: np.matmul(a, b)
    tf.transpose = lambda a, perm=None: np.transpose(a, perm)
    tf.gather_nd = _gather_nd
    tf.multiply = lambda a, b: np.multiply(a, b)
    tf.reshape = lambda a, shape: np.reshape(a, shape)
    tf.squeeze = lambda a, axis=None: np.squeeze(a, axis=axis)
    tf.expand_dims = lambda a, axis: np.expand_dims(a, axis)
    tf.one_hot = _one_hot
    tf.reduce_mean = lambda a, axis=None: np.mean(np.asarray(a), axis=axis)
    tf.nn = types.SimpleNamespace(softmax=_softmax)

    keras = types.ModuleType("keras")
    kl = types.ModuleType("keras.layers")
    for cls in (Layer, Conv1D, Dense, BatchNormalization, LeakyReLU, Activation, Dropout, SpatialDropout1D,
                Concatenate, MaxPool1D):
        setattr(kl, cls.__name__, cls)
    kl.Input = Input
    kr = types.ModuleType("keras.regularizers")
    kr.l2 = lambda *_a, **_k: None
    keras.layers, keras.regularizers = kl, kr
    keras.Model, keras.Layer = Model, Layer
    return {"tensorflow": tf, "keras": keras, "keras.layers": kl, "keras.regularizers": kr}


def install():
    """Register the stand-in modules (refuses to shadow a real TensorFlow / Keras)."""
    for name, mod in _make_modules().items():
        present = sys.modules.get(name)
        if present is not None and not getattr(present, "__genomad_shim__", False):
            raise RuntimeError(f"a real {name!r} module is loaded; the shim is only for images without it")
        mod.__genomad_shim__ = True
        sys.modules[name] = mod


def new_session(provider, dtype=np.float32):
    """Reset the per-class layer counters and install the weight provider
    ``provider(layer_class_name, index_of_that_class, weight_name) -> array``."""
    _STATE["provider"], _STATE["counters"], _STATE["dtype"] = provider, {}, dtype


# Our flat weight schema (genomad_amd/weights.py) -> (layer class, n-th instance, Keras weight name),
# in the order create_classifier() instantiates the layers (model.py:14-45, igloo.py:44-83).
_CONV = ["conv1", "conv2", "conv3"]
_DENSE = ["enc_dense", "head_dense", "out_dense"]
_BN = ["enc_bn", "head_bn"]
_IGLOO = ["iglooA", "iglooB"]
_BN_NAMES = {"gamma": "gamma", "beta": "beta", "moving_mean": "mean", "moving_variance": "var"}


def schema_provider(weights):
    def provider(cls, index, name):
        if cls == "Conv1D":
            return weights[f"{_CONV[index]}_{name}"]
        if cls == "Dense":
            return weights[f"{_DENSE[index]}_{name}"]
        if cls == "BatchNormalization":
            return weights[f"{_BN[index]}_{_BN_NAMES[name]}"]
        if cls == "IGLOO1D_kernel":
            return weights[f"{_IGLOO[index]}_{'patches' if name == 'random_patches' else name}"]
        raise KeyError((cls, index, name))
    return provider
