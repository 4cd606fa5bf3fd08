"""MI355X-native drop-in for the `mug` package of Keytoyze/Mug-Diffusion (sampling path only).

Put `mug-diffusion_amd/` ahead of the reference checkout on sys.path: the YAML
`target:` strings (`mug.diffusion.diffusion.DDPM`, ...) then resolve to these classes,
whose forward passes run in libmugd.so (hand-written HIP for gfx950).

Modules of the reference's `mug` package that are NOT on the hot path and are not provided
here (mug.data.utils, mug.data.dataset, mug.firststage.losses, mug.lr_scheduler, ...) keep
resolving to the reference's files: every other `mug/` directory found on sys.path is
appended to this package's search path (and likewise for the sub-packages)."""
import os as _os
import sys as _sys


def _fallthrough(package_path, *sub):
    """Other `mug/<sub>` directories on sys.path, in order, for modules this package does not shadow."""
    mine = _os.path.realpath(package_path[0])
    extra = []
    for p in _sys.path:
        cand = _os.path.realpath(_os.path.join(p or ".", "mug", *sub))
        if cand != mine and _os.path.isdir(cand) and cand not in extra and cand not in package_path:
            extra.append(cand)
    return extra


__path__ = list(__path__) + _fallthrough(__path__)
