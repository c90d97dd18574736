"""Synthetic Pixie-like pixel matrices: a 32-component mixture in
channel space, ~10 % zeroed entries, rows sum-normalised and divided by a per-channel
99.9 % vector -- the value distribution PixelSOMCluster feeds pyFlowSOM
(/root/reference/src/ark/phenotyping/cluster_helpers.py:218, 242-246)."""
import numpy as np
import torch

N_COMPONENTS = 32


def centers(c: int, seed: int = 7) -> np.ndarray:
    return np.random.RandomState(seed).uniform(0.0, 1.0, size=(N_COMPONENTS, c))


def make_fov_numpy(n: int, c: int, seed: int, dtype=np.float32) -> np.ndarray:
    """One FOV's retained-pixel matrix [n, c] on the host (tests; CPU baseline sample)."""
    rs = np.random.RandomState(seed)
    cen = centers(c)
    z = rs.randint(0, N_COMPONENTS, size=n)
    x = np.maximum(0.0, cen[z] + 0.05 * rs.standard_normal((n, c)))
    x[rs.uniform(size=(n, c)) < 0.10] = 0.0
    rsum = x.sum(axis=1, keepdims=True)
    rsum[rsum == 0] = 1.0
    x = x / rsum
    q = np.quantile(x, 0.999, axis=0)
    q[q == 0] = 1.0
    return np.ascontiguousarray((x / q).astype(dtype))


def make_fov_torch(n: int, c: int, seed: int, device, dtype=torch.float32) -> torch.Tensor:
    """Same distribution generated directly in HBM (bench: no PCIe in the way)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cen = torch.from_numpy(centers(c)).to(device=device, dtype=torch.float32)
    z = torch.randint(0, N_COMPONENTS, (n,), generator=g, device=device)
    x = cen[z] + 0.05 * torch.randn((n, c), generator=g, device=device)
    x.clamp_(min=0.0)
    x.mul_((torch.rand((n, c), generator=g, device=device) >= 0.10).to(x.dtype))
    rsum = x.sum(dim=1, keepdim=True)
    rsum[rsum == 0] = 1.0
    x.div_(rsum)
    # per-channel 99.9 % value from a 1M-row sample (exact value is irrelevant to the workload)
    samp = x[: min(n, 1 << 20)]
    q = torch.quantile(samp, 0.999, dim=0)
    q[q == 0] = 1.0
    x.div_(q)
    return x.to(dtype).contiguous()
