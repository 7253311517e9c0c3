"""Projector — same class and state-dict keys as the reference's `image/model/modules.py:12-55`
(`model.0` 1x1 conv without bias, `model.1` BatchNorm2d, `model.2` ReLU, `model.3` 1x1 conv with bias).
Parameter container: the arithmetic is part of `image_encoder.ImageEncodeFn`."""
from typing import Callable, Optional

import torch
import torch.nn as nn


class MLP(nn.Module):
    """Maps image patch embeddings into the joint projection space.

    :param input_dim: Input embedding feature size
    :param hidden_dim: Hidden layer size
    :param output_dim: Output projection size
    :param use_1x1_convs: Use 1x1 conv kernels (the only form the BioViL image model instantiates, model.py:103-104).
    """

    def __init__(self, input_dim: int, output_dim: int, hidden_dim: Optional[int] = None,
                 use_1x1_convs: bool = False) -> None:
        super().__init__()
        if not use_1x1_convs or hidden_dim is None:
            raise NotImplementedError("only the 1x1-conv projector with a hidden layer (ImageModel's use) is implemented")
        self.output_dim = output_dim
        self.input_dim = input_dim
        self.model = nn.Sequential(
            nn.Conv2d(in_channels=input_dim, out_channels=hidden_dim, kernel_size=1, bias=False),
            nn.BatchNorm2d(hidden_dim),
            nn.ReLU(inplace=True),
            nn.Conv2d(in_channels=hidden_dim, out_channels=output_dim, kernel_size=1, bias=True))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise RuntimeError("MLP projector is a parameter container; run it through ImageModel")
