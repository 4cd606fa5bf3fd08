"""Base class of the networks whose forward pass runs in libmugd.so."""
import torch
import torch.nn as nn

from mug.model import paramtree


class NativeModule(nn.Module):
    """Holds the reference-compatible parameters as ordinary torch Parameters (so state dicts,
    .cuda(), .to() work as usual) and lends their device memory to a native network instance.
    There is no torch implementation of the forward pass: the module must live on the GPU."""

    _kind = None

    def _setup(self, spec, cfg, s4_init=None):
        paramtree.register_spec(self, spec)
        paramtree.init_spec(self, spec, s4_init)
        self._spec_keys = [k for k, _, _ in spec]
        self._cfg = dict(cfg)
        self._native = None
        self._fp = None

    def _make_native(self, lib):
        raise NotImplementedError

    def native(self):
        from mug._native import get_lib
        lib = get_lib()
        dev = next(self.parameters()).device
        if dev != lib.device:
            raise RuntimeError("%s lives on %s but libmugd runs on %s: move the model there (model.cuda()); "
                               "there is no CPU implementation" % (type(self).__name__, dev, lib.device))
        if self._native is None or self._native.lib is not lib:
            self._native = self._make_native(lib)
            self._fp = None
        fp = paramtree.fingerprint(self)
        if fp != self._fp:
            tensors = dict(self.named_parameters())
            tensors.update(dict(self.named_buffers()))
            self._native.set_params(tensors)
            self._fp = fp
        return self._native
