"""Geometry helpers (reference: generators/math_utils_torch.py:8-26)."""
import torch


def transform_vectors(matrix: torch.Tensor, vectors4: torch.Tensor) -> torch.Tensor:
    return torch.matmul(vectors4, matrix.T)


def normalize_vecs(vectors: torch.Tensor) -> torch.Tensor:
    return vectors / (torch.norm(vectors, dim=-1, keepdim=True))


def torch_dot(x: torch.Tensor, y: torch.Tensor):
    return (x * y).sum(-1)
