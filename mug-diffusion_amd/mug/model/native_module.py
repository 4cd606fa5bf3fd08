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
        self._tensors = None          # cached [(name, tensor)]: walking the module tree costs more than the forward launches

    def _make_native(self, lib):
        raise NotImplementedError

    def _apply(self, fn, *args, **kwargs):      # .to() / .cuda() / .float(): parameters may be re-created
        self._tensors = None
        return super()._apply(fn, *args, **kwargs)

    def _tensor_list(self):
        if self._tensors is None:
            self._tensors = list(self.named_parameters()) + list(self.named_buffers())
        return self._tensors

    def _fingerprint(self):
        """(data_ptr, in-place version) of every tensor: detects .to(), load_state_dict, optimiser steps and other in-place
        updates since the values were last handed to the library (which packs weights / bakes S4 kernels from them)."""
        return hash(tuple((t.data_ptr(), t._version) for _, t in self._tensor_list()))

    def native(self):
        from mug._native import get_lib
        lib = get_lib()
        dev = self._tensor_list()[0][1].device
        if dev != lib.device:
            raise RuntimeError("%s lives on %s but libmugd runs on %s: move the model there (model.cuda()); "
                               "there is no CPU implementation" % (type(self).__name__, dev, lib.device))
        if self._native is None or self._native.lib is not lib:
            self._native = self._make_native(lib)
            self._fp = None
        fp = self._fingerprint()
        if fp != self._fp:
            self._tensors = None                  # re-walk once: a parameter object may have been replaced
            self._native.set_params(dict(self._tensor_list()))
            self._fp = self._fingerprint()
        return self._native
