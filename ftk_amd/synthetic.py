"""Synthetic benchmark inputs generated directly in HBM with torch (device memory plumbing).

Closed forms of the reference generators include/ftk/ndarray/synthetic.hh (woven 11-45, double_gyre 130-217,
moving_extremum 332-354) with the stream parameterisation of include/ftk/ndarray/stream.hh:1444-1567.  Arrays are laid out
like the reference's ndarray (first index fastest): torch shape (DH, DW) / (DD, DH, DW), vectors (..., ncomp).
moving_extremum with dyadic parameters is exact in FP64, hence bit-identical to the reference's pow(x, 2.0) loop
(tests/test_gpu_parity.py checks that against the oracle); the transcendental cases agree to rounding."""
import math


def moving_extremum_params(dims):
    """dyadic centre/velocity: V is an exact multiple of 1/4 -> nbits = 8, no int64 overflow"""
    off = (0.25, 0.375, 0.125)
    dv = (0.5, 0.25, 0.125)
    n = len(dims)
    return [dims[a] / 2 + off[a] for a in range(n)], list(dv[:n])


def moving_extremum(dims, k, x0, dirv, torch, device):
    n = len(dims)
    ax = [(torch.arange(dims[a], dtype=torch.float64, device=device) - (x0[a] + dirv[a] * float(k))) ** 2 for a in range(n)]
    # reference order: d = 0; d += (x-xc)^2; d += (y-yc)^2; [d += (z-zc)^2]   (synthetic.hh:343-349)
    if n == 2:
        return (ax[0][None, :] + ax[1][:, None]).contiguous()
    return ((ax[0][None, None, :] + ax[1][None, :, None]) + ax[2][:, None, None]).contiguous()


def woven(dims, k, nt, torch, device, scaling_factor=15.0):
    DW, DH = dims
    t = 0.0 if nt == 1 else float(k) / (nt - 1)
    x = ((torch.arange(DW, dtype=torch.float64, device=device) / (DW - 1)) - 0.5) * scaling_factor
    y = ((torch.arange(DH, dtype=torch.float64, device=device) / (DH - 1)) - 0.5) * scaling_factor
    X, Y = x[None, :], y[:, None]
    ct, st = math.cos(t), math.sin(t)
    return (torch.cos(X * ct - Y * st) * torch.sin(X * st + Y * ct)).contiguous()


def double_gyre(dims, k, torch, device, A=0.1, omega=2 * math.pi, eps=0.25, time_scale=0.1):
    DW, DH = dims
    time = k * time_scale
    x = (torch.arange(DW, dtype=torch.float64, device=device) / (DW - 1)) * 2
    y = (torch.arange(DH, dtype=torch.float64, device=device) / (DH - 1))
    X, Y = x[None, :], y[:, None]
    a = eps * math.sin(omega * time)
    b = 1 - 2 * eps * math.sin(omega * time)
    f = a * X * X + b * X
    dfdx = 2 * a * X + b
    u = -math.pi * A * torch.sin(math.pi * f) * torch.cos(math.pi * Y)
    v = math.pi * A * torch.cos(math.pi * f) * torch.sin(math.pi * Y) * dfdx
    return torch.stack([u.expand(DH, DW), v.expand(DH, DW)], dim=-1).contiguous()


def generate(case, dims, k, nt, torch, device):
    if case == "moving_extremum_3d_overflow":
        # the reference's own defaults for the velocity and a centre 1e-7 off the lattice: tiny non-zero gradient
        # components push nbits to 21, the quantised magnitudes reach D * 2^20 and the int64 determinants wrap almost everywhere
        n = len(dims)
        x0 = [dims[a] / 2 + (a + 1) * 1e-7 for a in range(n)]
        return moving_extremum(dims, k, x0, [0.1, 0.11, 0.1][:n], torch, device)
    if case in ("moving_extremum_2d", "moving_extremum_3d"):
        x0, dv = moving_extremum_params(dims)
        return moving_extremum(dims, k, x0, dv, torch, device)
    if case == "woven":
        return woven(dims, k, nt, torch, device)
    if case == "double_gyre":
        return double_gyre(dims, k, torch, device)
    raise ValueError(case)
