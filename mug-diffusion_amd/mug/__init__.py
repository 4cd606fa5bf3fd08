"""MI355X-native drop-in for the `mug` package of Keytoyze/Mug-Diffusion (sampling path only).

Put `mug-diffusion_amd/` ahead of the reference checkout on sys.path: the YAML
`target:` strings (`mug.diffusion.diffusion.DDPM`, ...) then resolve to these classes,
whose forward passes run in libmugd.so (hand-written HIP for gfx950)."""
