"""
oracle/ -- TEST INFRASTRUCTURE ONLY.

A CPU (NumPy) restatement of the reference algorithm for the GRAPE propagation
hot path, plus a hand-derived reverse-mode
adjoint that plays the role HIPS autograd plays in the reference.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package, and only as the checker / the timed CPU
baseline.  Nothing under `qoc_amd/` imports it; the product path has no CPU
fallback.

Parity status
-------------
* forward (states, costs, expm, Magnus, interpolation, clip/strip/slap):
  PINNED -- checked against golden vectors minted by importing the reference's
  own forward path in the build container (tools/gen_golden.py, fixtures in
  tests/golden/*.npz) and against the reference's analytic known answers
  (iSWAP, cost KATs).
* gradients: the reference pins NO gradient value in its tests and its AD
  engine (HIPS autograd, unpinned third party) is absent from this image, so
  the gradient oracle is pinned by (1) Richardson central differences of the
  REFERENCE forward and (2) an independent reverse-mode AD (PyTorch CPU
  complex128) over the same op sequence; both are stored in tests/golden.
"""
