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
* ``keras_shim`` / ``reference_harness`` – no restatement: they run the reference's files where
  they lie (numba / tensorflow / keras replaced by minimal stand-ins).
* ``make_golden`` / ``make_golden_config2`` – generate the committed fixtures of ``tests/golden/`` from them (config 2: all
  10 000 windows; config 3: every 512th of its 1 048 576 windows).
* ``precision_study`` / ``precision_mixes`` – numpy emulation of MFMA operand formats (which arithmetic keeps the scores
  inside the tolerance, and with what margin); design aids, not checkers.

Pinning status
--------------
* Integer half (tokenizer / windowing / FASTA): PINNED.  The reference's own
  ``genomad.sequence`` module is executed in this container by
  ``oracle/reference_harness.py`` (numba.njit stubbed as identity) and its
  outputs are committed under ``tests/golden/`` by ``oracle/make_golden.py``.
* Floating-point half (IGLOO forward): PINNED TO THE REFERENCE'S OWN NETWORK CODE, with one
  declared gap.  TensorFlow/Keras (pyproject.toml:12,19, un-pinned) are not installed here, the
  reference has no tests or golden outputs, and the trained weights
  (genomad/data/nn_classifier.h5) are absent from the checkout (.MISSING_LARGE_BLOBS).  So the
  reference's ``genomad/neural_network/model.py`` and ``igloo.py`` are EXECUTED IN PLACE
  (``reference_harness.reference_classifier_scores``): ``create_classifier()`` builds its graph
  and ``IGLOO1D_kernel.call`` runs its transposes / gather_nd / reshapes / matmuls unmodified,
  over numpy stand-ins for the ~20 TF/Keras primitives it calls (``oracle/keras_shim.py``, each
  a few lines following the documented Keras semantics).  The oracle agrees with that graph to
  3e-15 (fp64) / 1.3e-6 (fp32); its scores are committed in tests/golden/forward_golden.npz
  (``scores_refgraph64/32``) and the oracle AND the device path are tested against them.
  The gap: the primitives themselves (causal Conv1D, BatchNormalization inference, MaxPool1D,
  softmax ...) are our numpy, not TensorFlow's kernels; they are additionally cross-checked
  against an independent torch-CPU implementation in ``tests/test_oracle.py``.  Scores are
  compared on seeded synthetic weights of the exact reference shapes.
"""
