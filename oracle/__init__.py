"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT.

A CPU restatement (PyTorch-CPU fp32 functional ops + numpy host math) of the
reference's sampling hot path: mel front-end -> wave
encoder -> DDIM loop over the 1-D U-Net -> VAE decode -> thresholded note grid.

Every function cites the reference file:line it follows.  The restatement is
pinned against the *real* reference (imported in the authoring container via
oracle/refimport.py) by oracle/gen_golden.py, which writes the fixtures under
tests/golden/.  The reference ships no tests / golden vectors of its own
(SURVEY.md D11), so those generated fixtures are the pin.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  The product package
(mug-diffusion_amd/) never imports it and has no CPU fallback.
"""
