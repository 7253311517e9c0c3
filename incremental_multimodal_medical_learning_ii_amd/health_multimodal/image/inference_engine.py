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
    """Inference-time operations on an image model."""

    def __init__(self, image_model: ImageModel, transform: Compose):
        assert isinstance(image_model, ImageModel), f"Expected an ImageModel, got {type(image_model)}"
        self.model = image_model
        self.transform = transform
        self.model.eval()
        self.resize_size, self.crop_size = infer_resize_params(self.transform.transforms)
        self.to = self.model.to

    def load_and_transform_input_image(self, image_path: Path, transform: Callable) -> Tuple[torch.Tensor, TypeShape2D]:
        """Read an image, apply the transform, add the batch dimension, move to the model's device."""
        image = load_image(image_path)
        size = tuple(image.size) if hasattr(image, "size") and not isinstance(image, torch.Tensor) else tuple(image.shape[-2:][::-1])
        device = next(self.model.parameters()).device
        transformed_image = transform(image).unsqueeze(0).to(device)
        return transformed_image, size

    @torch.no_grad()
    def get_projected_patch_embeddings(self, image_path: Path) -> Tuple[torch.Tensor, TypeShape2D]:
        """L2-normalised patch embeddings [h, w, feature_size] and the original (width, height)."""
        input_image, img_shape = self.load_and_transform_input_image(image_path, self.transform)
        projected_img_emb = self.model.get_patchwise_projected_embeddings(input_image, normalize=True)
        assert projected_img_emb.shape[0] == 1
        return projected_img_emb[0], img_shape

    @torch.no_grad()
    def get_projected_global_embedding(self, image_path: Path) -> torch.Tensor:
        """L2-normalised global image embedding [joint_feature_dim]."""
        input_image, _ = self.load_and_transform_input_image(image_path, self.transform)
        return self.get_projected_global_embedding_from_tensor(input_image)[0]

    @torch.no_grad()
    def get_projected_global_embedding_from_tensor(self, images: torch.Tensor) -> torch.Tensor:
        """Batched form: images [B,3,H,W] on the model's device -> L2-normalised [B, joint_feature_dim]."""
        emb = self.model.forward(images)
        assert emb.ndim == 2
        xhat, _ = K.l2norm_fwd(emb)
        return xhat
