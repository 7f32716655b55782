import torch


def is_inside(points: torch.tensor, range_per_dim: torch.tensor):
    """N bool: whether each N x d point lies inside the d x 2 (min, max) range, bounds inclusive
    (reference volume.py:4-10)."""
    lo, hi = range_per_dim[:, 0], range_per_dim[:, 1]
    return ((points >= lo) & (points <= hi)).all(dim=-1)
