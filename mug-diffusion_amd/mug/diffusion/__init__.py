from mug import _fallthrough

__path__ = list(__path__) + _fallthrough(__path__, "diffusion")      # non-hot-path modules keep resolving to the reference's files
