"""Small geometry helpers with the reference's names (generators/math_utils_torch.py:8-26)."""
import torch


def normalize_vecs(vectors: torch.Tensor) -> torch.Tensor:
    """Unit-length vectors along the last axis (plain division by the 2-norm, no epsilon -- like the reference)."""
    length = torch.norm(vectors, dim=-1, keepdim=True)
    return vectors / length


def transform_vectors(matrix: torch.Tensor, vectors4: torch.Tensor) -> torch.Tensor:
    """[N,M] row vectors times an [M,M] matrix applied from the left: returns (matrix @ v) for every row v."""
    return vectors4 @ matrix.transpose(0, 1)


def torch_dot(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    return torch.sum(x * y, dim=-1)
