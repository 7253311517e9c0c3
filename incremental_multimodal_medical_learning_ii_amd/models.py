"""Adapters — same classes, attribute layout and state-dict keys as the reference's `models.py:7-26`
(`layer` is an `nn.Sequential` of `nn.Linear`/`nn.ReLU`, so the pickled `image_adapter.pt` / `text_adapter.pt`
layout of `Trainer.save`, Trainer.py:1643-1648, is preserved).  `forward` runs the cxrk HIP kernels."""
import torch
import torch.nn as nn

from . import functional as Fh


class myMLP(nn.Module):
    def __init__(self):
        super(myMLP, self).__init__()
        self.layer = nn.Sequential(nn.Linear(128, 256), nn.ReLU(), nn.Linear(256, 128))

    def forward(self, x):
        lead = x.shape[:-1]
        y = Fh.mlp_adapter(x.reshape(-1, x.shape[-1]), self.layer[0].weight, self.layer[0].bias,
                           self.layer[2].weight, self.layer[2].bias)
        return y.reshape(*lead, y.shape[-1])


class myLinearModel(nn.Module):
    def __init__(self):
        super(myLinearModel, self).__init__()
        self.layer = nn.Sequential(nn.Linear(128, 128))

    def forward(self, x):
        lead = x.shape[:-1]
        y = Fh.linear(x.reshape(-1, x.shape[-1]), self.layer[0].weight, self.layer[0].bias)
        return y.reshape(*lead, y.shape[-1])
