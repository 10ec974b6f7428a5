"""CPU oracle for the TemporalAlignNet hot path -- TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU fp32 restatement of the reference's algorithm (model/tfm_model.py,
model/tan_model.py, train/loss.py, the train() step of train/main.py and
eval/eval_zeroshot_align.py:test_alignment_htm).  It is pinned against golden vectors that
were produced by importing the real reference in the build container
(tests/golden/make_goldens.py).

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
package, and only as the checker / the timed CPU baseline.  Nothing under
temporalalignnet_amd/ imports it; the product path fails loudly if the HIP library is missing.
"""
