import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu")


def _gpu_explicitly_requested(config) -> bool:
    expr = config.getoption("-m") or ""
    return "gpu" in expr and "not gpu" not in expr


@pytest.fixture(scope="session")
def synth_weights():
    from genomad_amd import synthetic
    return synthetic.synth_weights()


@pytest.fixture(scope="session")
def engine(request, synth_weights):
    """One NNEngine on device 0 with the synthetic weights.  No CPU fallback: with `-m gpu`
    a missing library or device is a failure; without it the GPU tests are skipped."""
    from genomad_amd import _lib
    from genomad_amd.engine import NNEngine
    try:
        eng = NNEngine(0, synth_weights)
    except Exception as exc:  # noqa: BLE001
        if _gpu_explicitly_requested(request.config):
            raise
        pytest.skip(f"no usable gfx950 device: {exc}")
    yield eng
    eng.close()


@pytest.fixture
def kmer_tables(engine):
    """The session engine with the k-mer tables of "f16x3tk" built (156 GB; 0.2 .. 6 s the first time, nothing when they are there:
    function scope, because a test may drop them to make room for a subprocess); tests that need them are skipped on a device that
    cannot hold them."""
    if not engine.build_kmer_tables():
        pytest.skip("the device cannot hold the k-mer tables (156 GB + workspaces)")
    return engine


def need_tables(request, prec):
    """parametrised tests: build / require the k-mer tables only for the arithmetic that reads them"""
    if prec == "f16x3tk":
        request.getfixturevalue("kmer_tables")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
