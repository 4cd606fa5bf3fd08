"""TEST-ONLY stub: contract -> torch.einsum."""
import torch


def contract(expr, *ops, **kw):
    return torch.einsum(expr, *ops)


def contract_expression(*a, **k):
    raise NotImplementedError
