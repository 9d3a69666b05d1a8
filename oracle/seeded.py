"""ORACLE tooling (test infrastructure, NOT product code): deterministic parameter fill for whole-model fixtures.

A 240 M-parameter pSp or a 30 M-parameter StyleGAN2 Generator cannot be committed as a fixture, so both sides -- the imported
reference in ``oracle/gen_golden.py`` and the HIP modules / the CPU restatement in ``tests/`` -- fill every parameter and
buffer from ONE seeded ``torch.Generator`` by walking ``state_dict()`` in sorted key order, and the fixture stores only the
seed, the inputs, the outputs and a handful of gradients.  ``torch.randn`` on a CPU generator is bit-reproducible across
machines for a given torch build, and both sides draw the same shapes in the same order, so the parameters are identical."""
from __future__ import annotations

import math

import torch


def _rule(key: str, v: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    name = key.rsplit(".", 1)[-1]
    if key.endswith("running_mean"):
        return 0.1 * r
    if key.endswith("running_var"):
        return 0.5 + torch.sigmoid(r)
    if key.startswith("noises.") or key.endswith("input.input"):
        return r
    if key.endswith("modulation.bias"):
        return 1.0 + 0.1 * r
    if key.endswith("noise.weight"):
        return 0.3 * r
    if key.startswith("style.") and name == "weight":          # mapping network, lr_mul 0.01 (stylegan2/model.py:146)
        return r * 100.0
    if "latent_avg" in key:
        return 0.5 * r
    if v.ndim >= 4 and v.shape[0] == 1 and v.ndim == 5:        # ModulatedConv2d.weight [1, out, in, k, k]: its own init is randn
        return r
    if key.endswith("modulation.weight") or key.endswith("linear.weight"):  # EqualLinear with lr_mul 1: randn
        return r
    if v.ndim == 4:                                            # plain convolution: unit-gain fan-in scaling
        fan_in = v.shape[1] * v.shape[2] * v.shape[3]
        gain = 4.0 if ".conv.weight" in key and "attention" in key else 1.0
        return r * (gain / math.sqrt(fan_in))
    if v.ndim == 2:
        return r / math.sqrt(v.shape[1])
    if name == "weight":                                       # BatchNorm scale / PReLU slope: (0.25, 0.75)
        return 0.25 + 0.5 * torch.sigmoid(r)
    if name == "bias" and v.ndim == 4:                         # ToRGB bias [1, 3, 1, 1]
        return 0.2 * r
    return 0.1 * r                                             # biases


def seeded_fill_(module: torch.nn.Module, seed: int, prefix: str = "") -> None:
    """in-place; FIR kernels (``*.kernel`` buffers), LPIPS' ``mean`` / ``std`` constants and integer buffers keep their values.  ``prefix`` lets a sub-module be
    filled exactly as it would be as part of the parent (keys are sorted WITH the prefix applied)."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    seen = set()
    with torch.no_grad():
        for k in sorted(sd, key=lambda k: prefix + k):
            v = sd[k]
            if not v.is_floating_point() or k.endswith(".kernel") or k.rsplit(".", 1)[-1] in ("mean", "std") or v.data_ptr() in seen:
                continue
            seen.add(v.data_ptr())
            r = torch.randn(v.shape, generator=g, dtype=torch.float32)
            v.copy_(_rule(prefix + k, v, r).to(v.dtype))


def seeded_tensor(shape, seed: int, scale: float = 1.0) -> torch.Tensor:
    return torch.randn(tuple(shape), generator=torch.Generator().manual_seed(seed)) * scale


def grad_digest(g: torch.Tensor, keep: int = 4096) -> dict:
    """what a fixture keeps of a large gradient tensor: its sum, absolute sum, largest entry and a fixed strided sample"""
    f = g.detach().reshape(-1)
    step = max(1, f.numel() // keep)
    return dict(sum=f.double().sum().float(), abs_sum=f.abs().double().sum().float(), max=f.abs().max(), sample=f[::step].clone(), step=torch.tensor(step))


def check_digest(g: torch.Tensor, d: dict, tol: float, name: str = "") -> None:
    """a gradient against the digest a fixture keeps of it: every sampled entry within tol * max|g|, the signed sum within
    tol * sum|g|, the absolute sum within tol relative"""
    f = g.detach().reshape(-1).cpu().float()
    step = int(d["step"])
    mx = float(d["max"])
    err = float((f[::step] - d["sample"]).abs().max())
    assert err <= tol * mx + 1e-12, f"{name}: sampled entries off by {err:.3e} (max|g| {mx:.3e}, tol {tol})"
    asum = float(d["abs_sum"])
    assert abs(float(f.double().sum()) - float(d["sum"])) <= tol * asum + 1e-12, f"{name}: sum {float(f.double().sum()):.6e} vs {float(d['sum']):.6e}"
    assert abs(float(f.abs().double().sum()) - asum) <= tol * asum + 1e-12, f"{name}: abs-sum {float(f.abs().double().sum()):.6e} vs {asum:.6e}"


def digest_error(g: torch.Tensor, d64: dict) -> float:
    """largest deviation of the sampled entries from the float64 digest, relative to the tensor's largest entry"""
    f = g.detach().reshape(-1).cpu().float()
    return float((f[::int(d64["step"])] - d64["sample"]).abs().max()) / (float(d64["max"]) + 1e-30)


def check_adjudicated(grads: dict, d32: dict, d64: dict, floor: float = 2e-3, what: str = "", med_factor: float = 2.0, p90_factor: float = 3.0) -> None:
    """gradients (name -> tensor) against the reference's FLOAT64 digests, bounded by the error DISTRIBUTION of the reference's
    OWN fp32 run against the same float64 values (errors relative to each tensor's largest entry):
        median <= med_factor (2) x the reference's median,  90th percentile <= p90_factor (3) x the reference's,  worst <= max(4 x the reference's worst, floor)
    (with fewer than 8 tensors: every tensor <= max(4 x the reference's worst, floor)).
    Why a distribution and not a per-tensor bound: end-to-end gradients of these networks are decided at the 1e-3 .. 1e-2 level by
    chance events that differ between ANY two fp32 evaluations -- PReLU / LeakyReLU kink flips of pre-activations below the forward
    rounding error (batch 2, 4 x 4 maps in the style heads: one flipped element is 1 / 32 of a bias gradient) and cancellation in
    sums of millions of signed terms (noise weights, BatchNorm gradients).  The reference's fp32 run itself is 2e-4 (median) to 2e-2
    (worst) away from float64 on the whole-pSp fixture, and which tensor is hit is chance; the HIP path differs run to run by up to
    2e-2 on single tensors (fp32 atomics in split reductions move forward values in the last bit, which flips other kinks).  A
    structural error (wrong latent index, noise order, stride) is O(1) on the affected tensors and fails all three bounds."""
    names = [n for n, d in d64.items() if float(d["max"]) > 1e-20]
    for n in names:
        assert n in grads and grads[n] is not None, f"{what}: no gradient for {n}"
    ref = sorted(float((d32[n]["sample"] - d64[n]["sample"]).abs().max()) / float(d64[n]["max"]) for n in names)
    err = sorted((digest_error(grads[n], d64[n]), n) for n in names)
    q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
    worst_lim = max(4 * ref[-1], floor)
    msg = (f"{what}: {len(err)} tensors; error vs float64 median {q(err, .5)[0]:.2e} / p90 {q(err, .9)[0]:.2e} / worst {err[-1][0]:.2e} ({err[-1][1]}); "
           f"reference fp32: {q(ref, .5):.2e} / {q(ref, .9):.2e} / {ref[-1]:.2e}")
    print(msg)
    assert err[-1][0] <= worst_lim, msg
    if len(err) >= 8:
        assert q(err, .5)[0] <= med_factor * q(ref, .5) + 1e-6, msg
        assert q(err, .9)[0] <= p90_factor * q(ref, .9) + 1e-6, msg


def as_digest(t: torch.Tensor) -> dict:
    """a fully stored tensor in digest form (every entry sampled), so that check_adjudicated takes it"""
    f = t.detach().reshape(-1).float()
    return dict(sum=f.double().sum().float(), abs_sum=f.abs().double().sum().float(), max=f.abs().max(), sample=f, step=torch.tensor(1))


def criteria_inputs(seed: int):
    """x, y, ref, y_hat in [-1, 1] ([2, 3, 256, 256]) and a rectangular mask [2, 256, 256]: the inputs of the pSp criteria fixture"""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(2, 3, 256, 256, generator=g) * 2 - 1
    y = torch.rand(2, 3, 256, 256, generator=g) * 2 - 1
    rf = torch.rand(2, 3, 256, 256, generator=g) * 2 - 1
    yh = (y + 0.3 * torch.randn(2, 3, 256, 256, generator=g)).clamp(-1, 1)
    mask = torch.zeros(2, 256, 256)
    mask[0, 120:230, 60:200] = 1
    mask[1, 100:240, 40:180] = 1
    return x, y, rf, yh, mask
