"""Inference transform for chest X-rays on plain tensors (the reference builds the same pipeline from torchvision
transforms, `image/data/transforms.py:12-70`: Resize -> CenterCrop -> ToTensor -> ExpandChannels; NO mean/std
normalisation).  torchvision is not a dependency here; file decoding stays on the host and is outside the hot path."""
from typing import Callable, Sequence, Tuple

import torch
import torch.nn.functional as F


class ExpandChannels:
    """[1,H,W] -> [3,H,W] by repetition (reference `transforms.py:12-38`)."""

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        if data.shape[0] != 1:
            raise ValueError(f"Expected input of shape [1, H, W], found {data.shape}")
        return torch.repeat_interleave(data, 3, dim=0)


class Resize:
    def __init__(self, size: int):
        self.size = size

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        _, h, w = x.shape
        s = self.size / min(h, w)
        nh, nw = max(1, round(h * s)), max(1, round(w * s))
        return F.interpolate(x[None], size=(nh, nw), mode="bilinear", align_corners=False, antialias=True)[0]


class CenterCrop:
    def __init__(self, size: int):
        self.size = size

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        _, h, w = x.shape
        t, l = max(0, (h - self.size) // 2), max(0, (w - self.size) // 2)
        return x[:, t:t + self.size, l:l + self.size]


class ToTensor:
    """PIL image / uint8 array / float tensor -> float32 [1,H,W] in [0,1]."""

    def __call__(self, img) -> torch.Tensor:
        if isinstance(img, torch.Tensor):
            x = img
        else:
            import numpy as np
            x = torch.from_numpy(np.asarray(img))
        if x.dim() == 2:
            x = x[None]
        if x.dtype == torch.uint8:
            x = x.float() / 255.0
        return x.float()


class Compose:
    def __init__(self, transforms: Sequence[Callable]):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def create_chest_xray_transform_for_inference(resize: int, center_crop_size: int) -> Compose:
    """Resize, centre-crop, scale to [0,1], replicate to 3 channels (reference `transforms.py:41-52`)."""
    return Compose([ToTensor(), Resize(resize), CenterCrop(center_crop_size), ExpandChannels()])


def infer_resize_params(val_img_transforms: Sequence[Callable]) -> Tuple[int, int]:
    """Resize and crop sizes of a transform pipeline (reference `transforms.py:55-70`)."""
    resize = crop = None
    for t in val_img_transforms:
        if isinstance(t, Resize):
            resize = t.size
        elif isinstance(t, CenterCrop):
            crop = t.size
    return resize, crop
