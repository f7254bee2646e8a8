"""Run the reference's own integer code (tokenizer, windowing, FASTA reader) in place.

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.

The reference package cannot be imported normally in this image (tensorflow,
numba, rich_click, xgboost ... are absent and genomad/__init__.py:5-14 imports
every module).  ``genomad/sequence.py`` however only needs ``numba.njit`` (a
decorator, sequence.py:7,170) and ``genomad.utils`` (rich + numpy, both present).
So we register

* a bare ``genomad`` namespace module whose ``__path__`` points at the reference
  checkout (so ``genomad/__init__.py`` is never executed), and
* a stub ``numba`` whose ``njit`` is the identity,

and import ``genomad.sequence`` unmodified.  Nothing is copied: the reference's
files are executed where they lie.  ``/root/reference`` only exists in the build
container, so callers must guard with :func:`available`.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("GENOMAD_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "genomad", "sequence.py"))


def load_reference_sequence():
    """Return the reference's ``genomad.sequence`` module (tokenize_dna, seq_windows,
    read_fasta, Sequence) executed from /root/reference."""
    if not available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    if "genomad" not in sys.modules or not hasattr(sys.modules["genomad"], "__path__"):
        pkg = types.ModuleType("genomad")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "genomad")]
        sys.modules["genomad"] = pkg
    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")
        nb.njit = lambda f: f
        sys.modules["numba"] = nb
    return importlib.import_module("genomad.sequence")


def load_reference_utils():
    load_reference_sequence()
    return importlib.import_module("genomad.utils")


def load_reference_network():
    """Return the reference's ``genomad.neural_network`` package (model.py + igloo.py, executed
    from /root/reference) on top of the numpy stand-ins of ``oracle/keras_shim.py``."""
    load_reference_sequence()          # registers the bare ``genomad`` namespace package
    from oracle import keras_shim
    keras_shim.install()
    return importlib.import_module("genomad.neural_network")


def reference_classifier_scores(tokens, weights, dtype=None):
    """Scores of the reference's own ``create_classifier()`` graph (model.py:34-45) for (n,5997)
    tokens, with its layers' weights taken from our flat schema.  dtype float32 (as the reference
    runs) or float64."""
    import numpy as np
    from oracle import keras_shim
    nn = load_reference_network()
    keras_shim.new_session(keras_shim.schema_provider(weights), dtype or np.float32)
    model = nn.create_classifier()
    return model.predict(np.asarray(tokens, dtype=np.int64), verbose=0)
