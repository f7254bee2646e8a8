"""CPU oracle for the geNomad nn-classification hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``genomad_amd/`` imports this package.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.

What it restates (reference paths are relative to /root/reference):

* ``sequence_oracle``  – genomad/sequence.py:96-121 (read_fasta), :150-167
  (seq_windows), :170-193 (tokenize_dna) and the window filter / pad rule of
  genomad/modules/nn_classification.py:54-82.
* ``igloo_oracle``     – genomad/neural_network/model.py:9-45 and
  genomad/neural_network/igloo.py:30-83, :117-217 (forward pass only).

Pinning status
--------------
* Integer half (tokenizer / windowing / FASTA): PINNED.  The reference's own
  ``genomad.sequence`` module is executed in this container by
  ``oracle/reference_harness.py`` (numba.njit stubbed as identity) and its
  outputs are committed under ``tests/golden/`` by ``oracle/make_golden.py``.
* Floating-point half (IGLOO forward): PARITY UNPINNED.  The arithmetic lives in
  TensorFlow/Keras (pyproject.toml:12,19, un-pinned, not installed here), the
  reference has no tests or golden outputs, and the trained weights
  (genomad/data/nn_classifier.h5) are absent from the checkout
  (.MISSING_LARGE_BLOBS).  The restatement follows the reference's op sequence
  literally (``igloo_kernel_literal``) and is cross-checked against an
  independent torch-CPU implementation of conv1d/max_pool1d/softmax in
  ``tests/test_oracle.py``; scores are compared on seeded synthetic weights of
  the exact reference shapes.
"""
