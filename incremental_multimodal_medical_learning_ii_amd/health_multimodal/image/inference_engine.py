"""`ImageInferenceEngine` — same surface as the reference's `image/inference_engine.py:20-87`.  (The reference's
copy still reads `.projected_global_embedding` from an output object its edited `ImageModel.forward` no longer
returns — SURVEY.md F10; here the tensor returned by `forward` is used.)"""
from pathlib import Path
from typing import Callable, Tuple

import torch

from ... import kernels as K
from .data.io import load_image
from .data.transforms import Compose, infer_resize_params
from .model.model import ImageModel

TypeShape2D = Tuple[int, int]


class ImageInferenceEngine:
    """Image side of the zero-shot pipeline: file -> transform -> `ImageModel` -> L2-normalised embedding(s)."""

    def __init__(self, image_model: ImageModel, transform: Compose):
        if not isinstance(image_model, ImageModel):
            raise AssertionError(f"Expected an ImageModel, got {type(image_model)}")
        self.model = image_model.eval()
        self.transform = transform
        self.to = self.model.to
        # what the transform pipeline does to the image geometry; the similarity-map code undoes it (vlp/inference_engine.py)
        self.resize_size, self.crop_size = infer_resize_params(transform.transforms)

    def _device(self) -> torch.device:
        return next(self.model.parameters()).device

    def load_and_transform_input_image(self, image_path: Path, transform: Callable) -> Tuple[torch.Tensor, TypeShape2D]:
        """-> (a [1, C, H, W] batch on the model's device, the (width, height) the file had before the transform)."""
        image = load_image(image_path)
        if isinstance(image, torch.Tensor):
            original = (int(image.shape[-1]), int(image.shape[-2]))
        else:
            original = tuple(image.size)
        batch = transform(image)[None].to(self._device())
        return batch, original

    @torch.no_grad()
    def get_projected_patch_embeddings(self, image_path: Path) -> Tuple[torch.Tensor, TypeShape2D]:
        """Grid of L2-normalised patch embeddings [h, w, joint_feature_dim] of one image, and its original (width, height)."""
        batch, original = self.load_and_transform_input_image(image_path, self.transform)
        grid = self.model.get_patchwise_projected_embeddings(batch, normalize=True)
        assert grid.shape[0] == 1
        return grid[0], original

    @torch.no_grad()
    def get_projected_global_embedding(self, image_path: Path) -> torch.Tensor:
        """L2-normalised global embedding [joint_feature_dim] of one image."""
        batch, _ = self.load_and_transform_input_image(image_path, self.transform)
        return self.get_projected_global_embedding_from_tensor(batch)[0]

    @torch.no_grad()
    def get_projected_global_embedding_from_tensor(self, images: torch.Tensor) -> torch.Tensor:
        """Batched form used by the zero-shot evaluation: [B, 3, H, W] on the model's device -> L2-normalised [B, joint_feature_dim]."""
        emb = self.model.forward(images)
        assert emb.ndim == 2
        return K.l2norm_fwd(emb)[0]
