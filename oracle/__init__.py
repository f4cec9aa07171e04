"""CPU oracle for the FitHuBERT distillation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``fithubert_amd/`` imports this package.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline -- never as the thing shipped.

Contents
  student_ref.py   plain-torch fp32 restatement of the reference student
                   (modules/model.py:420-552, modules/module.py:24-102,266-358,
                   520-582,649-661)
  loss_ref.py      restatement of W2V2Distil.calculate_loss (train.py:236-378)
  teacher_ref.py   plain-torch HuBERT / wav2vec2 encoder with fairseq parameter
                   names + the TeacherWrapper contract (utils/utils.py:51-99)
  train_step_ref.py  one full CPU training step (the bench's cpu_baseline)
  gen_golden.py    (container only) imports /root/reference under
                   oracle/_standin to pin the restatement and to write
                   tests/golden/*.npz
  _standin/        minimal stand-in for the 14 fairseq symbols the reference
                   student imports (fairseq is not installed and there is no
                   network).  Used by gen_golden.py only; never by tests that
                   run on the GPU box.

Parity pinning: the reference has no tests and no golden vectors.
The restatement is pinned against the *unmodified* reference modules/ and
train.calculate_loss executed in this container (gen_golden.py), and the
fairseq-side semantics (MultiheadAttention, LayerNorm, GroupNorm, SamePad) that
the stand-in re-derives from memory are cross-checked against HuggingFace
``transformers.HubertModel`` (an independent implementation of the same
architecture) in tests/test_oracle.py.
"""
