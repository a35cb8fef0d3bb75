"""CPU oracle for the ocrs-models detection / recognition train-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker / reported baseline -- never as the
thing that is measured or shipped.  The product path (``ocrs_models_amd``)
raises if its HIP library is missing; it never falls back to this package.

What it is: a functional (parameter-dict based) restatement, in stock PyTorch
CPU operators and plain numpy loops, of

* ``ocrs_models/models.py:7-143``   DetectionModel  -> :mod:`oracle.detection`
* ``ocrs_models/models.py:146-268`` RecognitionModel -> :mod:`oracle.recognition`
* ``ocrs_models/train_detection.py:225-263`` balanced BCE -> :mod:`oracle.losses`
* ``torch.nn.CTCLoss`` as called at ``ocrs_models/train_rec.py:104,121`` -> :mod:`oracle.ctc`
* ``ocrs_models/train_rec.py:20-82,220-304`` and ``ocrs_models/datasets/util.py:27-35,113-177``
  (decode / collate / alphabet) -> :mod:`oracle.text`
* Adam / clip_grad_norm_ as used at ``train_detection.py:378``, ``train_rec.py:148,381`` -> :mod:`oracle.optim`

Pinning: the reference has no tests or golden vectors of its own (SURVEY.md
section 4), so the oracle is pinned against outputs of the imported reference
itself, generated in the build container by ``tools/gen_goldens.py`` and
committed as data under ``tests/golden/`` (``tests/test_oracle_golden.py``).
"""
