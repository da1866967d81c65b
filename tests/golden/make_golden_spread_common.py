"""Constants and helpers shared by make_golden_spread.py (imports the reference; build container only) and the test that checks the
oracle against its fixture (tests/test_oracle_golden.py; runs anywhere)."""
import torch

B, S, NCLS = 2, 1024, 8


def projection_vector(name, numel):
    """Fixed pseudo-random direction per tensor, seeded by the tensor's name and size."""
    g = torch.Generator().manual_seed(sum(name.encode()) * 7919 + numel)
    return torch.randn(numel, generator=g, dtype=torch.float64)


def summary(vals):
    v = sorted(vals)
    n = len(v)
    return dict(tensors=n, past_1e3=sum(x > 1e-3 for x in v), median=v[n // 2], p90=v[int(0.9 * n)], max=v[-1])
