"""TEST INFRASTRUCTURE.  Makes the read-only reference at /root/reference
importable in the authoring container (it needs pytorch_lightning, omegaconf,
opt_einsum, librosa, audioread, soundfile -- none installed; six tiny stubs in
oracle/refstubs stand in).  Used ONLY by oracle/gen_golden.py to produce the
fixtures under tests/golden/.  /root/reference does not exist on the GPU box,
so nothing under tests/ -m gpu, bench.py or smoke() imports this module."""
import os
import sys

REF = os.environ.get("MUG_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "mug"))


def activate():
    if not available():
        raise RuntimeError("reference checkout not present at %s" % REF)
    here = os.path.dirname(os.path.abspath(__file__))
    stubs = os.path.join(here, "refstubs")
    # drop any product 'mug' package already imported / on the path
    for k in [k for k in sys.modules if k == "mug" or k.startswith("mug.")]:
        del sys.modules[k]
    sys.path[:] = [p for p in sys.path if not p.rstrip("/").endswith("mug-diffusion_amd")]
    import torch  # import BEFORE the stubs are visible: torch probes for opt_einsum itself
    sys.path.insert(0, REF)
    sys.path.insert(0, stubs)
