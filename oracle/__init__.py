"""CPU oracle for the separation hot path -- TEST INFRASTRUCTURE ONLY.

This package is a numpy restatement of the arithmetic the reference
(Totoketchup/Adaptive-MultiSpeaker-Separation, a TF-1.x graph) performs on its
training/inference hot path.  It exists so the hand-written HIP kernels can be
checked; it is never part of the product:

  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import it;
  * nothing under ``adaptive-multispeaker-separation_amd/`` imports it, and the
    product raises if ``libams_hip.so`` is missing rather than falling back here.

PARITY UNPINNED.  The reference is Python-2 + TensorFlow 1.4 and cannot be
imported in the build container; TensorFlow itself (requirements.txt:6,11:
``tensorflow_gpu==1.4.0`` / ``tensorflow==1.5.0rc1``) is an un-vendored
third-party dependency and the reference ships no tests, golden vectors or
fixtures for this path.  Every function below cites the
reference file:line it restates and follows the documented TF-1.x op semantics
(SURVEY.md Appendix A); ``tests/test_oracle_*.py`` cross-check each one against
an independently written torch-CPU / scipy formulation.  The golden vectors in
``tests/golden`` are produced by this oracle (``tests/golden/make_golden.py``),
not by TensorFlow.

All functions take/return numpy arrays and are dtype-generic: float64 for the
parity oracle, float32 for the timed CPU baseline.
"""
from . import front, stft, blstm, dense, dpcl, l41, kmeans, separate, losses, optim, step, recipes  # noqa: F401
