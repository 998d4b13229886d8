"""ORACLE - TEST INFRASTRUCTURE ONLY.

A CPU restatement (plain PyTorch-CPU fp32 / numpy / C) of the reference's
algorithm for the NAS inner loop, used as the *checker* for the HIP path:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here;
  * the product package (``nas-segm-pytorch_amd/``) never does, and has no CPU
    fallback of its own.

Parity pinning.  The reference's own golden vectors (tests/precomputed/*.ckpt,
tests/test_inference.py:160-169) need pretrained weights from unreachable URLs,
so they pin nothing here.  Instead this oracle is pinned against outputs of the
reference itself: ``tests/golden/make_golden.py`` imports the reference from
/root/reference (build container only), runs it on seeded inputs and commits
the vectors under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks
every oracle function against them.  The arithmetic underneath (conv, BN,
interpolate, pooling, log-softmax, NLL) is third-party ``torch`` in the
reference (README.md:49, "torch>=1.0", not vendored); the oracle calls the same
``torch.nn.functional`` entry points on CPU, i.e. "torch 2.10 CPU semantics".
The one piece absent from the reference - the berHu loss of BASELINE config 5 -
is "parity unpinned" (restated from Laina et al. 2016, see oracle/losses.py).
"""
