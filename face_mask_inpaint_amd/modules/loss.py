"""Host-side mirror of modules/loss.py: VGGLoss (loss.py:16-65) and GANOptimizer (loss.py:68-144) with the
reference's signatures and return values; dice_* helpers belong to the mask-detector trainer and are out of scope.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as FF
from ..weights import packed
from .pluralistic_model import base_function
from .pluralistic_model.external_function import GANLoss, run_conv

VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M"]  # vgg16.features[:23] has no 5th stage


def _vgg16_features(width_div: int = 1) -> nn.Sequential:
    """Architecture of torchvision.models.vgg16().features[:24] (cfg 'D'); used when torchvision is absent.
    Weights are then random (no network access for the ImageNet checkpoint): SURVEY.md section 8c."""
    layers, c_in = [], 3
    for v in VGG_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c_in, v // width_div, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            c_in = v // width_div
    return nn.Sequential(*layers)


class VGGLoss(torch.nn.Module):
    def __init__(self, width_div: int = 1):
        super().__init__()
        try:  # the reference's source of weights (loss.py:21)
            import torchvision  # type: ignore

            features = torchvision.models.vgg16(pretrained=True).features
        except Exception:
            features = _vgg16_features(width_div)
        blocks = [features[:4].eval(), features[4:9].eval(), features[9:16].eval(), features[16:23].eval()]
        for bl in blocks:
            for p in bl.parameters():
                p.requires_grad = False
        self.blocks = torch.nn.ModuleList(blocks)
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))
        self._cache = {}

    def _prepare(self):
        """frozen weights: pack once, re-pack when a weight tensor is replaced or modified in place"""
        convs = [m for bl in self.blocks for m in bl if isinstance(m, nn.Conv2d)]
        key = tuple((c.weight.data_ptr(), c.weight._version) for c in convs)
        if self._cache.get("key") != key:
            with torch.no_grad():
                pws = FF.prepare_weights([(c.weight, None, None) for c in convs])
            self._cache = {"key": key, "pws": pws}
        for c, pw in zip(convs, self._cache["pws"]):
            object.__setattr__(c, "_fmi_packed", pw)

    def _block(self, bi, x):
        mods = [m for m in self.blocks[bi] if isinstance(m, (nn.MaxPool2d, nn.Conv2d))]
        prev_conv = False
        for i, m in enumerate(mods):
            if isinstance(m, nn.MaxPool2d):
                x = FF.max_pool2(x)
                prev_conv = False
            else:
                # ReLU fused into the GEMM epilogue.  Between two convolutions of a block the ReLU's backward runs in the NEXT
                # convolution's adjoint epilogue (its input is this ReLU's output and it is the only consumer), not as a pass of its own
                next_conv = i + 1 < len(mods) and isinstance(mods[i + 1], nn.Conv2d)
                x = run_conv(m, x, act=FF.ACT_RELU, in_act=("mask", 0.0) if prev_conv else None, skip_act_bwd=next_conv)
                prev_conv = True
        return x

    def _input(self, img):
        x = FF.to_nhwc(img)
        n, h, w, c = x.shape
        oh, ow = (224, 224) if w > 224 else (h, w)  # loss.py:48-49 "Filter HQ"
        return FF.resize_bilinear(x, oh, ow, self.mean.view(3).contiguous(), self.std.view(3).contiguous())

    def forward(self, input, target, lossType="perceptual"):
        self._prepare()
        x = self._input(input)
        with torch.no_grad():
            y = self._input(target)
        loss = 0.0
        for i in range(len(self.blocks)):
            x = self._block(i, x)
            with torch.no_grad():
                y = self._block(i, y)
            n, h, w, c = x.shape
            dim = c * h * w
            if lossType == "perceptual":
                loss = loss + FF.l1_loss(x, y) / dim
            elif lossType == "style":
                gx = FF.gram_matrix(x.view(n, h * w, c))
                gy = FF.gram_matrix(y.view(n, h * w, c))
                loss = loss + FF.l1_loss(gx, gy) / (c * c * dim)
            elif lossType == "contextual" and i == 3:
                loss = loss + FF.contextual_loss(x.view(n, h * w, c), y.view(n, h * w, c)) / dim
        return loss


    def forward_multi(self, triples):
        """[(input, target, lossType), ...] -> [loss, ...], identical values to calling ``forward`` per triple, but the
        inputs share ONE VGG pass with gradient (batch k*N) and the targets ONE pass without (the three losses of
        GANOptimizer are 6 VGG passes of batch N in the reference, loss.py:84-95): larger GEMM M fills the GPU at the
        28x28 / 56x56 layers, and the launch count drops 3x."""
        self._prepare()
        k = len(triples)
        n = triples[0][0].shape[0]
        x = torch.cat([self._input(t[0]) for t in triples], 0)
        with torch.no_grad():
            y = torch.cat([self._input(t[1]) for t in triples], 0)
        losses = [0.0] * k
        for i in range(len(self.blocks)):
            x = self._block(i, x)
            with torch.no_grad():
                y = self._block(i, y)
            _, h, w, c = x.shape
            dim = c * h * w
            xparts, yparts = x.split(n, 0), y.split(n, 0)  # contiguous batch slices; the backward of split is one cat
            for j, (_, _, lossType) in enumerate(triples):
                xs, ys = xparts[j], yparts[j]
                if lossType == "perceptual":
                    losses[j] = losses[j] + FF.l1_loss(xs, ys) / dim
                elif lossType == "style":
                    gx = FF.gram_matrix(xs.reshape(n, h * w, c))
                    gy = FF.gram_matrix(ys.reshape(n, h * w, c))
                    losses[j] = losses[j] + FF.l1_loss(gx, gy) / (c * c * dim)
                elif lossType == "contextual" and i == 3:
                    losses[j] = losses[j] + FF.contextual_loss(xs.reshape(n, h * w, c), ys.reshape(n, h * w, c)) / dim
        return losses


class GANOptimizer(nn.Module):
    """loss.py:68-144.  ``__call__`` performs the generator step then the discriminator step and returns
    (D_loss, G_loss, perc_loss, style_loss, cx_loss) exactly like the reference."""

    def __init__(self, optimizer_D, optimizer_G, lambda_g=0.01, debug=False, vgg_width_div: int = 1):
        super().__init__()
        self.gan_loss = GANLoss("lsgan")
        self.vgg_loss = VGGLoss(vgg_width_div)
        self.debug = debug
        self.optimizer_D = optimizer_D
        self.optimizer_G = optimizer_G
        self.lambda_perc = 0.1
        self.lambda_style = 250
        self.lambda_cx = 1
        self.lambda_g = lambda_g
        self.early_d = None  # None: automatic (on when the optimisers are DataParallelOptimizer); True / False force it

    @staticmethod
    def _masked(img, mask, invert):
        return FF.to_nchw(FF.mask_mul(FF.to_nhwc(img), mask.contiguous(), invert))

    def perceptual_loss(self, gt_img, gen_img):
        return self.vgg_loss(gen_img, gt_img, lossType="perceptual")

    def style_loss(self, gen_img, src_img, src_mask):
        return self.vgg_loss(self._masked(gen_img, src_mask, True), src_img, lossType="style")  # "Yes inverse"

    def contextual_loss(self, gen_img, ref_img, src_mask):
        return self.vgg_loss(self._masked(gen_img, src_mask, False), self._masked(ref_img, src_mask, False), lossType="contextual")

    def discriminator_loss(self, netD, real, fake):
        D_real_loss = self.gan_loss(netD(real), True, True)
        D_fake_loss = self.gan_loss(netD(fake.detach()), False, True)
        return (D_real_loss + D_fake_loss) * 0.5

    def generator_loss(self, netD, real, fake, freeze=True):
        if freeze:
            base_function._freeze(netD)
        D_fake = netD(fake)
        loss_ad_g = self.gan_loss(D_fake, True, False) * self.lambda_g
        loss_l1_g = FF.l1_loss(FF.to_nhwc(fake), FF.to_nhwc(real))
        return loss_ad_g + loss_l1_g

    def __call__(self, discriminator, src_img, gt_img, ref_img, gen_img, src_mask):
        # The reference leaves D trainable here (freeze=False) and then discards the D gradients this backward
        # produces (optimizer_D.zero_grad() at loss.py:130).  They are not computed at all: D's parameters are
        # switched to requires_grad=False for the generator pass only -- same values everywhere, less work.
        d_params = [p for p in discriminator.parameters() if p.requires_grad]
        for p in d_params:
            p.requires_grad_(False)
        try:
            G_loss = self.generator_loss(discriminator, gt_img, gen_img, freeze=False)
        finally:
            for p in d_params:
                p.requires_grad_(True)
        perc, sty, cx = self.vgg_loss.forward_multi([
            (gen_img, gt_img, "perceptual"),                                                     # loss.py:84-85
            (self._masked(gen_img, src_mask, True), src_img, "style"),                           # loss.py:87-89 "Yes inverse"
            (self._masked(gen_img, src_mask, False), self._masked(ref_img, src_mask, False), "contextual")])  # loss.py:91-95
        perc_loss, style_loss, cx_loss = perc * self.lambda_perc, sty * self.lambda_style, cx * self.lambda_cx
        G_loss = G_loss + perc_loss + style_loss + cx_loss
        early_d = self.early_d if self.early_d is not None else hasattr(self.optimizer_D, "launch")
        if early_d:
            # Data-parallel schedule: the discriminator loss depends only on gen_img and on the pre-update D
            # weights, so its forward (same D call order gen -> gt -> gen.detach(), hence the same SpectralNorm u/v
            # sequence) and backward run BEFORE the generator backward; the D gradients then travel over xGMI while the
            # VGG dgrad at the head of G_loss.backward() runs, and the G buckets follow as the decoder/encoder gradients
            # complete.  Every tensor has the value it has in the reference order (loss.py:120-134).
            D_loss = self.discriminator_loss(discriminator, gt_img, gen_img)
            self.optimizer_D.zero_grad()
            D_loss.backward()
            launch = getattr(self.optimizer_D, "launch", None)
            if launch is not None:
                launch()
            self.optimizer_G.zero_grad()
            G_loss.backward()
            self.optimizer_G.step()
            self.optimizer_D.step()
            return D_loss, G_loss, perc_loss, style_loss, cx_loss
        self.optimizer_G.zero_grad()
        G_loss.backward()
        self.optimizer_G.step()
        D_loss = self.discriminator_loss(discriminator, gt_img, gen_img)
        self.optimizer_D.zero_grad()
        D_loss.backward()
        self.optimizer_D.step()
        return D_loss, G_loss, perc_loss, style_loss, cx_loss

    def calc_loss(self, discriminator, src_img, gt_img, ref_img, gen_img, src_mask):
        D_loss = self.discriminator_loss(discriminator, gt_img, gen_img)
        G_loss = self.generator_loss(discriminator, gt_img, gen_img, freeze=False)
        perc_loss = self.perceptual_loss(gt_img, gen_img) * self.lambda_perc
        style_loss = self.style_loss(gen_img, src_img, src_mask) * self.lambda_style
        cx_loss = self.contextual_loss(gen_img, ref_img, src_mask) * self.lambda_cx
        return D_loss, G_loss + perc_loss + style_loss + cx_loss
