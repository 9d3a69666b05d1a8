"""Host-side mirror of modules/psp/criteria/id_loss.py: IDLoss = 1 - <ArcFace(y_hat), ArcFace(y)> on the cropped, 112 x 112-pooled
face (id_loss.py:22-50).  The ArcFace weights (model_paths['ir_se50']) are download-only: when the file is absent the IR-SE50
backbone keeps its random initialisation (weight parity unpinned; the arithmetic is pinned by tests/golden/psp_criteria.pt)."""
from __future__ import annotations

import os

import torch
from torch import nn

from .... import functional as FF
from .. import model_paths
from ..encoders.model_irse import Backbone


class IDLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.facenet = Backbone(input_size=112, num_layers=50, drop_ratio=0.6, mode="ir_se")
        if os.path.isfile(model_paths["ir_se50"]):
            self.facenet.load_state_dict(torch.load(model_paths["ir_se50"], map_location="cpu", weights_only=True))
        else:
            print("IDLoss: %s not found -> keeping the random initialisation" % model_paths["ir_se50"])
        self.face_pool = torch.nn.AdaptiveAvgPool2d((112, 112))
        self.facenet.eval()
        # the reference leaves the ArcFace parameters trainable but hands them to no optimiser: their gradients are computed and
        # dropped; here they are not computed (values of everything that is used are identical)
        for p in self.facenet.parameters():
            p.requires_grad_(False)

    def train(self, mode=True):  # the reference builds IDLoss().eval() and never switches it back (criteria/__init__.py:32)
        return super().train(False)

    def extract_feats(self, x):
        x = FF.to_nhwc(x)[:, 35:223, 32:220, :].contiguous()  # crop interesting region (id_loss.py:23)
        x = FF.adaptive_avg_pool(x, 112, 112)
        return self.facenet.nhwc(x)

    def forward(self, y_hat, y, x):
        n_samples = x.shape[0]
        with torch.no_grad():  # eval-mode network, no batch-coupled op: x and y go through as one batch of 2N
            xy = self.extract_feats(torch.cat([x, y], dim=0))
            x_feats, y_feats = xy[:n_samples], xy[n_samples:]
        y_hat_feats = self.extract_feats(y_hat)
        # loss = mean_i (1 - <y_hat_i, y_i>) as one reduction; the per-sample dot products are only logged
        loss = 1.0 - FF.dot_all(y_hat_feats, y_feats, 1.0 / n_samples)
        if getattr(self, "defer_logs", False):  # no host synchronisation (HIP-graph capture): the improvement stays a device scalar
            with torch.no_grad():
                imp = ((y_hat_feats.detach() * y_feats).sum() - (y_feats * x_feats).sum()) / n_samples
            return loss, imp, None
        with torch.no_grad():
            dt = (y_hat_feats.detach() * y_feats).sum(1).tolist()
            di = (y_hat_feats.detach() * x_feats).sum(1).tolist()
            dv = (y_feats * x_feats).sum(1).tolist()
        id_logs = [{"diff_target": float(a), "diff_input": float(b), "diff_views": float(c)} for a, b, c in zip(dt, di, dv)]
        sim_improvement = sum(a - c for a, c in zip(dt, dv)) / n_samples
        return loss, sim_improvement, id_logs
