"""ORACLE (test infrastructure, NOT product code): CPU restatement of the pSp encoder side of the hot path
(SURVEY.md 8a rows B1, B9, B10): IR-SE bottlenecks, GradualStyleBlock, GradualStyleEncoder, pSp.forward and the offline
terms of pSpLoss.  Functional torch on a flat parameter dictionary with the reference's state_dict keys.  BatchNorm
running statistics are updated in ``P`` in place when ``training`` (torch's rule: momentum 0.1, unbiased variance).
Pinned by tests/golden/psp_ops.pt, produced by the reference's own modules (oracle/gen_golden.py: psp_fixtures).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import picnet_cpu as pic
from . import stylegan2_cpu as sg2

Params = Dict[str, torch.Tensor]


def batch_norm(P: Params, prefix: str, x: torch.Tensor, training: bool, eps: float = 1e-5, momentum: float = 0.1) -> torch.Tensor:
    """nn.BatchNorm2d as used at helpers.py:83,87,104,108 and psp_encoders.py:47"""
    w, b = P[prefix + ".weight"], P[prefix + ".bias"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        n = x.numel() // x.shape[1]
        with torch.no_grad():
            P[prefix + ".running_mean"] = (1 - momentum) * P[prefix + ".running_mean"] + momentum * mean.detach()
            P[prefix + ".running_var"] = (1 - momentum) * P[prefix + ".running_var"] + momentum * var.detach() * (n / max(n - 1, 1))
            P[prefix + ".num_batches_tracked"] = P[prefix + ".num_batches_tracked"] + 1
    else:
        mean, var = P[prefix + ".running_mean"], P[prefix + ".running_var"]
    xh = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + eps)
    return xh * w.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def prelu(P: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    a = P[prefix + ".weight"].view(1, -1, 1, 1)
    return torch.where(x >= 0, x, a * x)


def se_module(P: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """helpers.py:56-72"""
    s = x.mean(dim=(2, 3), keepdim=True)
    s = F.relu(F.conv2d(s, P[prefix + ".fc1.weight"]))
    s = torch.sigmoid(F.conv2d(s, P[prefix + ".fc2.weight"]))
    return x * s


def bottleneck(P: Params, prefix: str, x: torch.Tensor, stride: int, training: bool) -> torch.Tensor:
    """bottleneck_IR / bottleneck_IR_SE (helpers.py:75-119); the SE stage runs when its weights exist"""
    if prefix + ".shortcut_layer.0.weight" in P:
        sc = F.conv2d(x, P[prefix + ".shortcut_layer.0.weight"], stride=stride)
        sc = batch_norm(P, prefix + ".shortcut_layer.1", sc, training)
    else:
        sc = x[:, :, ::stride, ::stride]  # MaxPool2d(1, stride)
    r = batch_norm(P, prefix + ".res_layer.0", x, training)
    r = F.conv2d(r, P[prefix + ".res_layer.1.weight"], padding=1)
    r = prelu(P, prefix + ".res_layer.2", r)
    r = F.conv2d(r, P[prefix + ".res_layer.3.weight"], stride=stride, padding=1)
    r = batch_norm(P, prefix + ".res_layer.4", r, training)
    if prefix + ".res_layer.5.fc1.weight" in P:
        r = se_module(P, prefix + ".res_layer.5", r)
    return r + sc


def gradual_style_block(P: Params, prefix: str, x: torch.Tensor, out_c: int) -> torch.Tensor:
    """psp_encoders.py:13-36"""
    i = 0
    while prefix + ".convs.%d.weight" % i in P:
        x = F.leaky_relu(F.conv2d(x, P[prefix + ".convs.%d.weight" % i], P[prefix + ".convs.%d.bias" % i], stride=2, padding=1), 0.01)
        i += 2
    return sg2.equal_linear(P, prefix + ".linear", x.reshape(-1, out_c))


def body_strides(num_layers: int = 50):
    units = {50: (3, 4, 14, 3), 100: (3, 13, 30, 3), 152: (3, 8, 36, 3)}[num_layers]
    s = []
    for u in units:
        s += [2] + [1] * (u - 1)
    return s


def pyramid(P: Params, prefix: str, x: torch.Tensor, training: bool, strides, taps=(6, 20, 23)):
    x = F.conv2d(x, P[prefix + "input_layer.0.weight"], padding=1)
    x = prelu(P, prefix + "input_layer.2", batch_norm(P, prefix + "input_layer.1", x, training))
    out = []
    for i, s in enumerate(strides):
        x = bottleneck(P, prefix + "body.%d" % i, x, s, training)
        if i in taps:
            out.append(x)
    return out


def gradual_style_encoder(P: Params, prefix: str, x, ref=None, mask=None, n_styles=14, use_attention=True, training=True,
                          strides=None, taps=(6, 20, 23)) -> torch.Tensor:
    """psp_encoders.py:100-152; ``strides``/``taps`` default to the IR-50 body"""
    strides = body_strides(50) if strides is None else strides
    c1, c2, c3 = pyramid(P, prefix, x, training, strides, taps)
    if ref is not None:
        m = mask.unsqueeze(1)
        r1, r2, r3 = pyramid(P, prefix, ref, training, strides, taps)
        m3, m2, m1 = (pic.scale_img(m, r.shape[-2:]) for r in (r3, r2, r1))
        if use_attention:
            c3 = pic.example_guided_attention(P, prefix + "attention1", m3, c3, r3)
            c2 = pic.example_guided_attention(P, prefix + "attention2", m2, c2, r2)
        else:
            c3 = m3 * r3 + (1 - m3) * c3
            c2 = m2 * r2 + (1 - m2) * c2
        c1 = m1 * r1 + (1 - m1) * c1
    oc = P[prefix + "styles.0.linear.weight"].shape[0]
    lat = [gradual_style_block(P, prefix + "styles.%d" % j, c3, oc) for j in range(3)]

    def up_add(a, b):
        return F.interpolate(a, size=b.shape[-2:], mode="bilinear", align_corners=True) + b

    p2 = up_add(c3, F.conv2d(c2, P[prefix + "latlayer1.weight"], P[prefix + "latlayer1.bias"]))
    lat += [gradual_style_block(P, prefix + "styles.%d" % j, p2, oc) for j in range(3, 7)]
    p1 = up_add(p2, F.conv2d(c1, P[prefix + "latlayer2.weight"], P[prefix + "latlayer2.bias"]))
    lat += [gradual_style_block(P, prefix + "styles.%d" % j, p1, oc) for j in range(7, n_styles)]
    return torch.stack(lat, dim=1)


def psp_forward(P: Params, x, ref, mask, noises, size: int, latent_avg: Optional[torch.Tensor] = None, training=True, resize=True,
                use_attention=True):
    """psp.py:72-120 with explicit noise maps (randomize_noise draws them in the reference, stylegan2/model.py:486-492)"""
    n_styles = int(math.log(size, 2)) * 2 - 2
    codes = gradual_style_encoder(P, "encoder.", x, ref, mask, n_styles, use_attention, training)
    if latent_avg is not None:
        codes = codes + latent_avg.repeat(codes.shape[0], 1, 1)
    PD = {k[len("decoder."):]: v for k, v in P.items() if k.startswith("decoder.")}
    img = sg2.generator_forward(PD, codes, noises, size)
    if resize:
        img = F.adaptive_avg_pool2d(img, (256, 256))
    return img, codes


def w_norm(latent, latent_avg=None):
    """criteria/w_norm.py:5-14"""
    if latent_avg is not None:
        latent = latent - latent_avg
    return torch.sum(latent.norm(2, dim=(1, 2))) / latent.shape[0]


def psp_loss(y, y_hat, latent, latent_avg=None, ref=None, mask=None, l2_lambda=1.0, l2_lambda_ref=1.0, w_norm_lambda=0.0,
             start_from_latent_avg=True):
    """criteria/__init__.py:44-99 for the terms that enter ``loss`` with the LPIPS / ID lambdas at 0"""
    loss = 0.0
    m = mask.unsqueeze(1) if mask is not None else None
    if l2_lambda > 0:
        loss = loss + l2_lambda * (F.mse_loss(y_hat * (1 - m), y * (1 - m)) if m is not None else F.mse_loss(y_hat, y))
    if ref is not None and l2_lambda_ref > 0:
        loss = loss + l2_lambda_ref * F.mse_loss(y_hat * m, ref * m)
    if w_norm_lambda > 0 and latent_avg is not None:
        loss = loss + w_norm_lambda * w_norm(latent, latent_avg if start_from_latent_avg else None)
    return loss


# ---- LPIPS(alex) and the ArcFace identity loss (criteria/lpips/*, criteria/id_loss.py, encoders/model_irse.py) -------------
ALEX_LAYERS = (("conv", 0, 4, 2), ("relu",), ("pool",), ("conv", 3, 1, 2), ("relu",), ("pool",), ("conv", 6, 1, 1), ("relu",),
               ("conv", 8, 1, 1), ("relu",), ("conv", 10, 1, 1), ("relu",), ("pool",))
ALEX_TARGETS = (2, 5, 8, 10, 12)


def lpips_alex(P: Params, prefix: str, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """LPIPS.forward (lpips.py:30-36) with BaseNet.forward (networks.py:49-62) and normalize_activation (utils.py:6-8)"""
    def feats(t):
        t = (t - P[prefix + "net.mean"]) / P[prefix + "net.std"]
        out = []
        for i, op in enumerate(ALEX_LAYERS, 1):
            if op[0] == "conv":
                t = F.conv2d(t, P[f"{prefix}net.layers.{op[1]}.weight"], P[f"{prefix}net.layers.{op[1]}.bias"], stride=op[2], padding=op[3])
            elif op[0] == "relu":
                t = F.relu(t)
            else:
                t = F.max_pool2d(t, 3, 2)
            if i in ALEX_TARGETS:
                out.append(t / (torch.sqrt(torch.sum(t ** 2, dim=1, keepdim=True)) + 1e-10))
        return out

    res = [F.conv2d((a - b) ** 2, P[f"{prefix}lin.{j}.1.weight"]).mean((2, 3), True) for j, (a, b) in enumerate(zip(feats(x), feats(y)))]
    return torch.sum(torch.cat(res, 0)) / x.shape[0]


def arcface_backbone(P: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """Backbone(112, 50, 'ir_se').forward in eval mode (model_irse.py:39-43) + l2_norm (helpers.py:15-18)"""
    x = F.conv2d(x, P[prefix + "input_layer.0.weight"], padding=1)
    x = prelu(P, prefix + "input_layer.2", batch_norm(P, prefix + "input_layer.1", x, False))
    for i, s in enumerate(body_strides(50)):
        x = bottleneck(P, f"{prefix}body.{i}", x, s, False)
    x = batch_norm(P, prefix + "output_layer.0", x, False).flatten(1)
    x = F.linear(x, P[prefix + "output_layer.3.weight"], P[prefix + "output_layer.3.bias"])
    x = F.batch_norm(x, P[prefix + "output_layer.4.running_mean"], P[prefix + "output_layer.4.running_var"], P[prefix + "output_layer.4.weight"],
                     P[prefix + "output_layer.4.bias"], False, 0.1, 1e-5)
    return x / torch.norm(x, 2, 1, True)


def id_loss(P: Params, prefix: str, y_hat, y, x):
    """IDLoss.forward (id_loss.py:28-50): returns (loss, sim_improvement, [diff_target, diff_input, diff_views] per sample)"""
    ext = lambda t: arcface_backbone(P, prefix + "facenet.", F.adaptive_avg_pool2d(t[:, :, 35:223, 32:220], (112, 112)))
    xf, yf, hf = ext(x), ext(y).detach(), ext(y_hat)
    dt, di, dv = (hf * yf).sum(1), (hf * xf).sum(1), (yf * xf).sum(1)
    return (1 - dt).mean(), float((dt.detach() - dv.detach()).mean()), torch.stack([dt, di, dv], 1).detach()


def psp_loss_full(P: Params, x, y, y_hat, latent, latent_avg, ref, mask, a):
    """pSpLoss.__call__ (criteria/__init__.py:44-99) with every term; ``a`` = the lambdas.  NOTE the reference ASSIGNS the identity
    term (``loss = loss_id * id_lambda``, :56) and adds the rest; style / contextual are logged only"""
    m = mask.unsqueeze(1)
    loss = 0.0
    if a["id_lambda"] > 0:
        loss = id_loss(P, "id_loss.", y_hat, y, x)[0] * a["id_lambda"]
    if a["l2_lambda"] > 0:
        loss = loss + F.mse_loss(y_hat * (1 - m), y * (1 - m)) * a["l2_lambda"]
    if a["lpips_lambda"] > 0:
        loss = loss + lpips_alex(P, "lpips_loss.", y_hat * (1 - m), y * (1 - m)) * a["lpips_lambda"]
    if a["lpips_lambda_ref"] > 0:
        loss = loss + lpips_alex(P, "lpips_loss.", y_hat * m, ref * m) * a["lpips_lambda_ref"]
    if a["l2_lambda_ref"] > 0:
        loss = loss + F.mse_loss(y_hat * m, ref * m) * a["l2_lambda_ref"]
    if a["w_norm_lambda"] > 0:
        loss = loss + w_norm(latent, latent_avg) * a["w_norm_lambda"]
    return loss
