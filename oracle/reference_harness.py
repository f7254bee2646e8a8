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
