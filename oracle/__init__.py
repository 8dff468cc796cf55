"""CPU oracle for the noise-trajectory-search hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU (torch-CPU / numpy) restatement of the
reference algorithm (rvignav/diffusion-tts) for the path named in BASELINE.json.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it,
and only as the checker / reported CPU baseline -- never as the thing measured or shipped.
The product package (`diffusion_tts_amd`) never imports it and has no CPU fallback.

Parity pin: every function here is checked against outputs of the reference itself, produced
by importing the reference in the build container (tests/golden/make_golden.py, committed with
the vectors it wrote under tests/golden/*.npz).  The reference has no tests of its own for this
path, so those vectors are the pin.

All `file:line` citations are relative to the reference checkout (/root/reference).
"""
