"""Shared test helpers: layout conversion and error measures."""
import torch


def to_cl(x: torch.Tensor, dtype, device):
    """NCHW / NCDHW fp32 (CPU) -> channels-last contiguous tensor of `dtype` on `device`."""
    perm = [0] + list(range(2, x.dim())) + [1]
    return x.permute(*perm).contiguous().to(device=device, dtype=dtype)


def from_cl(y: torch.Tensor):
    """channels-last (any device / dtype) -> NCHW fp32 on the CPU."""
    perm = [0, y.dim() - 1] + list(range(1, y.dim() - 1))
    return y.float().cpu().permute(*perm).contiguous()


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).float()


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b|  (scale-relative maximum error)."""
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))
