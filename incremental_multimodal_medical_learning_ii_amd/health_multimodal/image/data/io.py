"""Image loading on the host (outside the hot path).  The reference reads JPEG/PNG/DICOM/NIfTI through PIL, pydicom
and SimpleITK (`image/data/io.py:17-71`); only PIL-readable files, `.npy` arrays and `.pt` tensors are handled here."""
from pathlib import Path

import numpy as np
import torch


def load_image(path: Path):
    path = Path(path)
    if path.suffix == ".npy":
        return torch.from_numpy(np.load(path, allow_pickle=False))
    if path.suffix == ".pt":
        return torch.load(path, map_location="cpu", weights_only=True)
    if path.suffix.lower() in (".jpg", ".jpeg", ".png"):
        from PIL import Image
        return Image.open(path).convert("L")
    raise ValueError(f"Image type not supported on this path, filename was: {path}")
