"""CPU oracle for the UNet2DS hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and there only as the checker / the timed CPU baseline.  The product path
(``deep_calcium_amd``) never imports this package and fails loudly when the
HIP library is missing.

PARITY STATUS: **parity unpinned by the reference for the network arithmetic**
(the arithmetic lives in Keras==2.0.6 / tensorflow-gpu==1.2.1,
/root/reference/requirements.txt:28,:67, neither vendored nor installable
here; the reference has no tests or golden tensors).  The arithmetic is pinned
instead by two independent restatements that must agree to 1e-10
(``unet_numpy`` float64 hand-derived backward vs ``unet_torch`` float64
autograd) plus analytic known-answer tests (tests/test_oracle.py).  The
host-side pieces (TTA table, batch generator, validation coordinates,
reflect-pad) ARE pinned: tests/golden/*.npz were produced by importing the
reference's own functions in the build container
(tests/golden/make_goldens.py).
"""
