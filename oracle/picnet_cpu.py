"""ORACLE (test infrastructure, NOT product code).

CPU restatement, in functional pure-torch form, of the reference's PICNet-ref
training hot path.  Every function takes a
flat parameter dictionary ``P`` (name -> tensor, the reference's ``state_dict``
keys) instead of ``nn.Module`` objects, so the same code can be driven by the
reference's own checkpoint/state_dict, by the HIP product's state_dict, or by
the committed golden fixtures.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file.  The product package
(``face_mask_inpaint_amd``) never does.

Parity pin: ``tests/golden/picnet_*.pt`` were produced by importing the
reference itself (``oracle/gen_golden.py``) and this restatement is checked
against them in ``tests/test_oracle_golden.py``.

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

LRELU_SLOPE = 0.1  # base_function.py:61 nn.LeakyReLU(0.1)


# ----------------------------------------------------------------------------
# A11  spectral norm (external_function.py:30-41,70-72)
# ----------------------------------------------------------------------------
def _l2n(v: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    # external_function.py:12-13
    return v / (v.norm() + eps)


def sn_weight(P: Params, prefix: str) -> torch.Tensor:
    """One power iteration on ``.data`` (state mutated in place, also under
    no_grad / eval), then ``W / sigma`` with sigma = u . (W v) differentiable
    through W only.  ``prefix`` names the wrapped conv, keys are
    ``prefix.weight_bar/_u/_v``."""
    w = P[prefix + ".weight_bar"]
    u = P[prefix + ".weight_u"]
    v = P[prefix + ".weight_v"]
    h = w.shape[0]
    # NOTE the reference rebinds ``u.data`` / ``v.data`` (no autograd version bump) and keeps the
    # Parameters themselves in the graph.  When the same conv runs twice before one backward
    # (D(real) then D(fake) in discriminator_loss, loss.py:97-107) the FIRST call's sigma-gradient
    # is therefore evaluated with the u/v left behind by the SECOND call.  Restated as-is.
    with torch.no_grad():
        wm = w.detach().reshape(h, -1)
        v.data = _l2n(torch.mv(wm.t(), u.data))
        u.data = _l2n(torch.mv(wm, v.data))
    sigma = u.dot(w.reshape(h, -1).mv(v))
    return w / sigma.expand_as(w)


def sn_conv(P: Params, prefix: str, x: torch.Tensor, stride: int = 1, padding: int = 0) -> torch.Tensor:
    """SpectralNorm(nn.Conv2d) forward; ``prefix`` is the SpectralNorm module."""
    w = sn_weight(P, prefix + ".module")
    return F.conv2d(x, w, P.get(prefix + ".module.bias"), stride=stride, padding=padding)


def sn_conv_transpose(P: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """SpectralNorm(nn.ConvTranspose2d(k3,s2,p1,op1)) (base_function.py:326-341)."""
    w = sn_weight(P, prefix + ".module")
    return F.conv_transpose2d(x, w, P.get(prefix + ".module.bias"), stride=2, padding=1, output_padding=1)


def lrelu(x: torch.Tensor, slope: float = LRELU_SLOPE) -> torch.Tensor:
    return F.leaky_relu(x, slope)


def inst_norm(P: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    # base_function.py:47 nn.InstanceNorm2d(affine=True) -> eps 1e-5, no running stats
    return F.instance_norm(x, weight=P[prefix + ".weight"], bias=P[prefix + ".bias"], eps=1e-5)


# ----------------------------------------------------------------------------
# A2  blocks (base_function.py:207-398)
# ----------------------------------------------------------------------------
def res_block(P: Params, prefix: str, x: torch.Tensor, sample: str = "none", slope: float = LRELU_SLOPE) -> torch.Tensor:
    """ResBlock with norm_layer=None (the only form on the hot path):
    model = act, conv1, act, conv2 ; shortcut = bypass(1x1) (base_function.py:242-268)."""
    h = sn_conv(P, prefix + ".conv1", lrelu(x, slope), padding=1)
    h = sn_conv(P, prefix + ".conv2", lrelu(h, slope), padding=1)
    s = sn_conv(P, prefix + ".bypass", x, padding=0)
    if sample == "down":
        return F.avg_pool2d(h, 2, 2) + F.avg_pool2d(s, 2, 2)
    if sample == "none":
        return h + s
    raise NotImplementedError(sample)


def res_block_encoder_optimized(P: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """base_function.py:271-305 with norm none: conv1, act, conv2, avgpool || avgpool, bypass."""
    h = sn_conv(P, prefix + ".conv1", x, padding=1)
    h = sn_conv(P, prefix + ".conv2", lrelu(h), padding=1)
    h = F.avg_pool2d(h, 2, 2)
    s = sn_conv(P, prefix + ".bypass", F.avg_pool2d(x, 2, 2), padding=0)
    return h + s


def res_block_decoder(P: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """base_function.py:308-364 with InstanceNorm(affine): IN, act, conv1, IN, act, convT || convT."""
    h = lrelu(inst_norm(P, prefix + ".model.0", x))
    h = sn_conv(P, prefix + ".conv1", h, padding=1)
    h = lrelu(inst_norm(P, prefix + ".model.3", h))
    h = sn_conv_transpose(P, prefix + ".conv2", h)
    s = sn_conv_transpose(P, prefix + ".bypass", x)
    return h + s


def output_block(P: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """base_function.py:367-398 with norm None: act, ReflectionPad2d(1), conv3x3, tanh."""
    h = F.pad(lrelu(x), (1, 1, 1, 1), mode="reflect")
    return torch.tanh(sn_conv(P, prefix + ".conv1", h, padding=0))


def self_attention_core(q: torch.Tensor, values: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """softmax(q^T q) applied to value maps: out = V . A^T
    (example_guided_att.py:15-33, base_function.py:429-437)."""
    n = q.shape[0]
    qf = q.reshape(n, q.shape[1], -1)
    att = torch.softmax(qf.permute(0, 2, 1) @ qf, dim=-1)
    outs = []
    for v in values:
        vf = v.reshape(n, v.shape[1], -1)
        outs.append((vf @ att.permute(0, 2, 1)).reshape(v.shape))
    return outs


def auto_attn(P: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """Auto_Attn.forward with pre=None (base_function.py:420-448); the attention map is discarded by callers."""
    q = F.conv2d(x, P[prefix + ".query_conv.weight"], P[prefix + ".query_conv.bias"])
    (o,) = self_attention_core(q, [x])
    return P[prefix + ".gamma"] * o + x


def example_guided_attention(P: Params, prefix: str, mask: torch.Tensor, src: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
    """example_guided_att.py:21-41."""
    q = F.conv2d(src, P[prefix + ".conv.weight"])
    src_att, ref_att = self_attention_core(q, [src, ref])
    flow = (1 - mask) * ref_att + mask * ref
    out = torch.cat([flow, src_att], dim=1)
    if prefix + ".out_conv.weight" in P:
        out = F.conv2d(out, P[prefix + ".out_conv.weight"], P[prefix + ".out_conv.bias"])
    return out


# ----------------------------------------------------------------------------
# networks (network.py:76-370)
# ----------------------------------------------------------------------------
def res_encoder(P: Params, prefix: str, img: torch.Tensor, encoder_type: str, layers: int = 5, L: int = 6, z_nc: int = 128):
    """ResEncoder.forward (network.py:137-178)."""
    out = res_block_encoder_optimized(P, prefix + ".block0", img)
    for i in range(layers - 1):
        out = res_block(P, f"{prefix}.encoder{i}", out, "none" if i % 2 == 0 else "down")
    enc = out
    if encoder_type == "src":
        for i in range(L):
            enc = res_block(P, f"{prefix}.infer_prior{i}", enc)
        o = res_block(P, prefix + ".prior", enc)
    else:
        o = res_block(P, prefix + ".posterior", enc)
    mu, std = torch.split(o, z_nc, dim=1)
    return [mu, F.softplus(std)], out


def get_z(src_dist, ref_dist, eps_p: torch.Tensor, eps_q: torch.Tensor, return_zq: bool = False) -> torch.Tensor:
    """ResGenerator.get_z (network.py:275-307) with the two standard-normal
    draws injected (rsample = mu + sigma*eps, posterior drawn first)."""
    p_mu, p_sigma = ref_dist
    q_mu, q_sigma = src_dist
    z_p = p_mu + p_sigma * eps_p
    z_q = q_mu + q_sigma * eps_q
    if return_zq:
        return z_q
    return torch.cat([z_q, z_p], dim=1)


def res_generator(P: Params, prefix: str, encoded: torch.Tensor, z: Optional[torch.Tensor], layers: int = 5, L: int = 0, use_attn: bool = True) -> torch.Tensor:
    """ResGenerator.forward (network.py:247-273); Auto_Attn's default nonlinearity
    is nn.LeakyReLU() but its ResBlock is never run (pre is None)."""
    if z is not None:
        f = res_block(P, prefix + ".generator", z)
        for i in range(L):
            f = res_block(P, f"{prefix}.generator{i}", f)
        out = encoded + f
    else:
        out = encoded
    output = None
    for i in range(layers):
        out = res_block_decoder(P, f"{prefix}.decoder{i}", out)
        if i == 1 and use_attn:
            out = auto_attn(P, f"{prefix}.attn{i}", out)
        if i > layers - 2:
            output = output_block(P, f"{prefix}.out{i}", out)
            out = torch.cat([out, output], dim=1)
    return output


def res_discriminator(P: Params, prefix: str, x: torch.Tensor, layers: int = 5, use_attn: bool = True) -> torch.Tensor:
    """ResDiscriminator.forward (network.py:360-370)."""
    pre = prefix + "." if prefix else ""
    out = res_block_encoder_optimized(P, pre + "block0", x)
    for i in range(layers - 1):
        if i == 2 and use_attn:
            out = auto_attn(P, f"{pre}attn{i}", out)
        out = res_block(P, f"{pre}encoder{i}", out, "down")
    out = res_block(P, pre + "block1", out)
    return sn_conv(P, pre + "conv", lrelu(out), padding=0)


# ----------------------------------------------------------------------------
# A1 / A7  mask prep, ReferenceFill.forward (model.py:10-12, 81-112)
# ----------------------------------------------------------------------------
def patch_discriminator(P: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """PatchDiscriminator.forward (network.py:373-430): model.0, 2, 4, ... are SpectralNorm 4x4 convs (stride 2, then two of stride 1),
    LeakyReLU(0.1) between them, none after the last"""
    idx = sorted({int(k[len(prefix) + len("model."):].split(".")[0]) for k in P if k.startswith(prefix + "model.") and k.endswith("weight_bar")})
    for j, i in enumerate(idx):
        stride = 2 if j < len(idx) - 2 else 1
        x = sn_conv(P, f"{prefix}model.{i}", x, stride=stride, padding=1)
        if j < len(idx) - 1:
            x = lrelu(x)
    return x


def binarise_mask(mask_i64: torch.Tensor) -> torch.Tensor:
    """train_reference_fill.py:340  (mask > 0).float()"""
    return (mask_i64 > 0).float()


def scale_img(img: torch.Tensor, size) -> torch.Tensor:
    return F.interpolate(img, size=size, mode="bilinear", align_corners=True)


def reference_fill_forward(P: Params, src: torch.Tensor, ref: torch.Tensor, src_mask: torch.Tensor,
                           eps_p: torch.Tensor, eps_q: torch.Tensor, *, enc_layers: int = 5, enc_L: int = 6,
                           enc_z_nc: int = 128, dec_layers: int = 5, dec_L: int = 0, out_size=(256, 256),
                           use_att: bool = True, resize: bool = True, no_prior: bool = False) -> torch.Tensor:
    """ReferenceFill.forward (model.py:81-112) incl. its ``use_att=False`` blend and the ``no_prior`` form of --old_model"""
    src_dist, src_feat = res_encoder(P, "src_encoder", src, "src", enc_layers, enc_L, enc_z_nc)
    ref_dist, ref_feat = res_encoder(P, "ref_encoder", ref, "ref", enc_layers, enc_L, enc_z_nc)
    m = scale_img(src_mask.unsqueeze(1), src_feat.shape[-2:])
    if use_att:
        enc = example_guided_attention(P, "attention", m, src_feat, ref_feat)
    else:
        enc = (1 - m) * src_feat + m * ref_feat
    z = None if no_prior else get_z(src_dist, ref_dist, eps_p, eps_q, return_zq=not use_att)
    img = res_generator(P, "decoder", enc, z, dec_layers, dec_L)
    if resize:
        img = scale_img(img, (218, 178)) if no_prior else F.adaptive_avg_pool2d(img, out_size)
    return img


# ----------------------------------------------------------------------------
# A9  VGG losses (loss.py:16-65, external_function.py:180-192,231-274)
# ----------------------------------------------------------------------------
VGG_BLOCKS = (  # torchvision vgg16.features indices kept by the slicing at loss.py:22-25
    (("conv", 0, 3, 64), ("conv", 2, 64, 64)),
    (("pool",), ("conv", 5, 64, 128), ("conv", 7, 128, 128)),
    (("pool",), ("conv", 10, 128, 256), ("conv", 12, 256, 256), ("conv", 14, 256, 256)),
    (("pool",), ("conv", 17, 256, 512), ("conv", 19, 512, 512), ("conv", 21, 512, 512)),
)
VGG_MEAN = (0.485, 0.456, 0.406)
VGG_STD = (0.229, 0.224, 0.225)


def vgg_block(P: Params, prefix: str, bi: int, x: torch.Tensor) -> torch.Tensor:
    for op in VGG_BLOCKS[bi]:
        if op[0] == "pool":
            x = F.max_pool2d(x, 2, 2)
        else:
            k = f"{prefix}blocks.{bi}.{op[1]}"
            x = F.relu(F.conv2d(x, P[k + ".weight"], P[k + ".bias"], padding=1))
    return x


def gram_matrix(x: torch.Tensor) -> torch.Tensor:
    n, c, h, w = x.shape
    f = x.reshape(n, c, h * w)
    return torch.bmm(f, f.transpose(1, 2)) / (c * h * w)


def style_loss(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    return F.l1_loss(gram_matrix(x), gram_matrix(y).detach())


def contextual_loss(x: torch.Tensor, y: torch.Tensor, h: float = 0.5) -> torch.Tensor:
    """external_function.py:231-274."""
    n, c = x.shape[:2]
    y_mu = y.mean(3).mean(2).mean(0).reshape(1, -1, 1, 1)
    xc = x - y_mu
    yc = y - y_mu
    xn = (xc / torch.norm(xc, p=2, dim=1, keepdim=True)).reshape(n, c, -1)
    yn = (yc / torch.norm(yc, p=2, dim=1, keepdim=True)).reshape(n, c, -1)
    d = 1 - torch.bmm(xn.transpose(1, 2), yn)
    d_min, _ = torch.min(d, dim=2, keepdim=True)
    w = torch.exp((1 - d / (d_min + 1e-5)) / h)
    cx_ij = w / torch.sum(w, dim=2, keepdim=True)
    cx = torch.mean(torch.max(cx_ij, dim=1)[0], dim=1)
    return torch.mean(-torch.log(cx + 1e-5))


def vgg_loss(P: Params, prefix: str, inp: torch.Tensor, tgt: torch.Tensor, loss_type: str) -> torch.Tensor:
    """VGGLoss.forward (loss.py:45-65)."""
    if inp.shape[-1] > 224:
        inp, tgt = scale_img(inp, [224, 224]), scale_img(tgt, [224, 224])
    mean = torch.tensor(VGG_MEAN, dtype=inp.dtype).view(1, 3, 1, 1)
    std = torch.tensor(VGG_STD, dtype=inp.dtype).view(1, 3, 1, 1)
    x = (inp - mean) / std
    y = (tgt - mean) / std
    loss = 0.0
    for i in range(4):
        x = vgg_block(P, prefix, i, x)
        y = vgg_block(P, prefix, i, y)
        dim = x.shape[1] * x.shape[2] * x.shape[3]
        if loss_type == "perceptual":
            loss = loss + F.l1_loss(x, y) / dim
        elif loss_type == "style":
            loss = loss + style_loss(x, y) / (x.shape[1] * x.shape[1] * dim)
        elif loss_type == "contextual" and i == 3:
            loss = loss + contextual_loss(x, y) / dim
    return loss


# ----------------------------------------------------------------------------
# A10  GAN losses + optimiser step (external_function.py:110-131, loss.py:84-134)
# ----------------------------------------------------------------------------
LAMBDA_G, LAMBDA_PERC, LAMBDA_STYLE, LAMBDA_CX = 0.01, 0.1, 250.0, 1.0


def lsgan(pred: torch.Tensor, target_is_real: bool) -> torch.Tensor:
    return F.mse_loss(pred, torch.full_like(pred, 1.0 if target_is_real else 0.0))


def generator_losses(PD: Params, PV: Params, src, gt, ref, gen, mask):
    """loss.py:109-124: returns (G_total, perc, style, cx)."""
    g = lsgan(res_discriminator(PD, "", gen), True) * LAMBDA_G + F.l1_loss(gen, gt)
    perc = vgg_loss(PV, "", gen, gt, "perceptual") * LAMBDA_PERC
    sty = vgg_loss(PV, "", gen * (1 - mask).unsqueeze(1), src, "style") * LAMBDA_STYLE
    cx = vgg_loss(PV, "", gen * mask.unsqueeze(1), ref * mask.unsqueeze(1), "contextual") * LAMBDA_CX
    return g + perc + sty + cx, perc, sty, cx


def discriminator_loss(PD: Params, real, fake):
    d_real = lsgan(res_discriminator(PD, "", real), True)
    d_fake = lsgan(res_discriminator(PD, "", fake.detach()), False)
    return (d_real + d_fake) * 0.5


def trainable(P: Params) -> List[torch.Tensor]:
    return [t for k, t in P.items() if t.requires_grad]


def prepare_params(sd: Dict[str, torch.Tensor], frozen: bool = False, dtype: torch.dtype = torch.float32) -> Params:
    """Clone a state_dict into leaf tensors (``dtype=torch.float64``: the same restatement as a double-precision adjudicator).  ``weight_u/_v`` are
    requires_grad=False Parameters in the reference (external_function.py:58-59).
    Aliased keys (``model.N`` / ``shortcut.N`` duplicates of conv1/conv2/bypass)
    are collapsed onto one tensor so that updates stay shared."""
    out: Params = {}
    seen: Dict[int, torch.Tensor] = {}
    for k, v in sd.items():
        key = v.data_ptr() if v.numel() else id(v)
        if key in seen and seen[key].shape == v.shape:
            out[k] = seen[key]
            continue
        t = v.detach().clone().to(dtype) if v.is_floating_point() else v.detach().clone()
        needs_grad = (not frozen) and not (k.endswith("weight_u") or k.endswith("weight_v"))
        t.requires_grad_(needs_grad)
        out[k] = t
        seen[key] = t
    return out


def unique_trainable(P: Params) -> List[torch.Tensor]:
    seen, out = set(), []
    for k, t in P.items():
        if t.requires_grad and id(t) not in seen:
            seen.add(id(t))
            out.append(t)
    return out


def train_step(PG: Params, PD: Params, PV: Params, opt_g: torch.optim.Optimizer, opt_d: torch.optim.Optimizer,
               src, gt, ref, mask_i64, eps_p, eps_q, **fwd_kw):
    """One iteration of train_reference_fill.py:331-346 + GANOptimizer.__call__ (loss.py:120-134)."""
    mask = binarise_mask(mask_i64)
    gen = reference_fill_forward(PG, src, ref, mask, eps_p, eps_q, **fwd_kw)
    g_loss, perc, sty, cx = generator_losses(PD, PV, src, gt, ref, gen, mask)
    opt_g.zero_grad()
    g_loss.backward()
    opt_g.step()
    d_loss = discriminator_loss(PD, gt, gen)
    opt_d.zero_grad()
    d_loss.backward()
    opt_d.step()
    return gen.detach(), d_loss.detach(), g_loss.detach(), perc.detach(), sty.detach(), cx.detach()


# ----------------------------------------------------------------------------
# synthetic inputs shared by bench.py's cpu_baseline leg and tests
# ----------------------------------------------------------------------------
def synthetic_batch(n: int, size: int = 256, seed: int = 1234, feat_hw: int = 32, z_nc: int = 128, bernoulli: bool = False):
    g = torch.Generator().manual_seed(seed)
    src = torch.rand(n, 3, size, size, generator=g)
    ref = torch.rand(n, 3, size, size, generator=g)
    gt = torch.rand(n, 3, size, size, generator=g)
    if bernoulli:
        mask = (torch.rand(n, size, size, generator=g) < 0.5).long() * 255
    else:
        yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
        s = size / 256.0
        mask = torch.zeros(n, size, size, dtype=torch.long)
        for i in range(n):
            cy = (176 + (torch.rand(1, generator=g).item() * 32 - 16)) * s
            cx = (128 + (torch.rand(1, generator=g).item() * 32 - 16)) * s
            ry = (56 + (torch.rand(1, generator=g).item() * 24 - 12)) * s
            rx = (80 + (torch.rand(1, generator=g).item() * 24 - 12)) * s
            inside = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
            mask[i][inside] = 255
    eps_p = torch.randn(n, z_nc, feat_hw, feat_hw, generator=g)
    eps_q = torch.randn(n, z_nc, feat_hw, feat_hw, generator=g)
    return src, ref, gt, mask, eps_p, eps_q
