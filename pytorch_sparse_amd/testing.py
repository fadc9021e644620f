"""Parametrisation helpers mirroring torch_sparse/testing.py (GPU only: there is no CPU path)."""
import torch

reductions = ['sum', 'add', 'mean', 'min', 'max']
dtypes = [torch.half, torch.bfloat16, torch.float, torch.double, torch.int, torch.long]
grad_dtypes = [torch.half, torch.bfloat16, torch.float, torch.double]
devices = [torch.device('cuda:0')] if torch.cuda.is_available() else []


def tensor(x, dtype, device):
    return None if x is None else torch.tensor(x, dtype=dtype, device=device)
