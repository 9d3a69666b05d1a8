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
    """in-place; FIR kernels (``*.kernel`` buffers) and integer buffers keep their values.  ``prefix`` lets a sub-module be
    filled exactly as it would be as part of the parent (keys are sorted WITH the prefix applied)."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    seen = set()
    with torch.no_grad():
        for k in sorted(sd, key=lambda k: prefix + k):
            v = sd[k]
            if not v.is_floating_point() or k.endswith(".kernel") or v.data_ptr() in seen:
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


def check_adjudicated(grads: dict, d32: dict, d64: dict, floor: float = 2e-3, factor: float = 3.0, what: str = "") -> None:
    """gradients (name -> tensor) against the reference's FLOAT64 digests, bounded by the reference's OWN fp32 error:
      err(t) <= max(factor * err_ref32(t), 2 * worst err_ref32 over the tensors of t's class, floor)
    with two classes -- reductions (scalars / vectors: bias, noise-weight, BatchNorm gradients, sums of millions of signed terms) and
    weight tensors -- and the median error over all tensors <= 2 x the reference's median.  (The reference's fp32 run itself is up
    to 2e-2 of max|g| away from float64 on the reductions and 2e-4 in the median, and which tensor is hit is chance: a fixed
    tolerance would be either vacuous or flaky.)"""
    names = [n for n, d in d64.items() if float(d["max"]) > 1e-20]
    for n in names:
        assert n in grads and grads[n] is not None, f"{what}: no gradient for {n}"
    ref = {n: float((d32[n]["sample"] - d64[n]["sample"]).abs().max()) / float(d64[n]["max"]) for n in names}
    cls = {n: int(grads[n].ndim <= 1) for n in names}
    cls_floor = {c: max([floor] + [2 * ref[n] for n in names if cls[n] == c]) for c in (0, 1)}
    errs = []
    for n in names:
        e = digest_error(grads[n], d64[n])
        lim = max(factor * ref[n], cls_floor[cls[n]])
        assert e <= lim, f"{what} {n}: {e:.3e} of max|g| from float64 > {lim:.3e} (reference fp32 on this tensor: {ref[n]:.3e})"
        errs.append(e)
    errs.sort()
    refs = sorted(ref.values())
    med, rmed = errs[len(errs) // 2], refs[len(refs) // 2]
    assert med <= 2 * rmed + 1e-6, f"{what}: median error {med:.3e} vs the reference's own {rmed:.3e}"
    print(f"{what}: {len(errs)} tensors, median error vs float64 {med:.2e} (reference fp32 {rmed:.2e}), worst {errs[-1]:.2e} (reference {refs[-1]:.2e}); "
          f"class bounds: weights {cls_floor[0]:.1e}, reductions {cls_floor[1]:.1e}")


def as_digest(t: torch.Tensor) -> dict:
    """a fully stored tensor in digest form (every entry sampled), so that check_adjudicated takes it"""
    f = t.detach().reshape(-1).float()
    return dict(sum=f.double().sum().float(), abs_sum=f.abs().double().sum().float(), max=f.abs().max(), sample=f, step=torch.tensor(1))
