"""`ImageTextInferenceEngine` — same surface as the reference's `vlp/inference_engine.py:21-160`.
Scores and the patch-wise similarity (`:104`) run on the HIP kernels; the gaussian smoothing (`:106-107`, scipy) and the
resize / NaN-pad to the original image size (`:111-155`) are host-side post-processing of a [h, w] map, as in the reference."""
import math
from pathlib import Path
from typing import Callable, List, Optional, Union

import numpy as np
import torch
import torch.nn.functional as F
from scipy import ndimage

from ... import functional as Fh
from ... import kernels as K
from ..image import ImageInferenceEngine
from ..text import TextInferenceEngine


class ImageTextInferenceEngine:
    """Joint image + text inference."""

    def __init__(self, image_inference_engine: ImageInferenceEngine, text_inference_engine: TextInferenceEngine) -> None:
        self.image_inference_engine = image_inference_engine
        self.text_inference_engine = text_inference_engine

    def _text_vector(self, query_text: Union[List[str], str]) -> torch.Tensor:
        query_text = [query_text] if isinstance(query_text, str) else query_text
        num_prompts = len(query_text)
        text_embedding = self.text_inference_engine.get_embeddings_from_prompt(query_text, normalize=False, verbose=False)
        assert text_embedding.shape[0] == num_prompts
        mean = K.group_mean_fwd(text_embedding, 1, num_prompts)       # average over prompts ...
        xhat, _ = K.l2norm_fwd(mean)                                  # ... then L2-normalise (:50-51)
        return xhat

    @torch.no_grad()
    def get_similarity_score_from_raw_data(self, image_path: Path, query_text: Union[List[str], str]) -> float:
        """Cosine similarity between an image and one or more strings (embeddings of several strings are averaged
        before L2-normalisation)."""
        assert not self.image_inference_engine.model.training
        assert not self.text_inference_engine.model.training
        image_embedding = self.image_inference_engine.get_projected_global_embedding(image_path)
        text_embedding = self._text_vector(query_text)
        out = torch.empty(1, 1, dtype=torch.float32, device=text_embedding.device)
        K.gemm(image_embedding[None].contiguous(), text_embedding, out, 1, 1, text_embedding.shape[1], False, True)
        return out.item()

    @torch.no_grad()
    def get_similarity_scores_from_tensors(self, images: torch.Tensor, class_prompts: List[List[str]]) -> torch.Tensor:
        """Batched zero-shot scores [B, C]: normalize(image_emb) @ normalize(mean prompt emb of class c)
        (the loop of `trash/lower_bound_mcs.py:79-117`, BASELINE config 1)."""
        assert not self.image_inference_engine.model.training
        assert not self.text_inference_engine.model.training
        img = self.image_inference_engine.get_projected_global_embedding_from_tensor(images)
        txt = torch.cat([self._text_vector(p) for p in class_prompts], dim=0)
        return Fh.similarity_logits(img, txt)

    @torch.no_grad()
    def get_similarity_map_from_raw_data(self, image_path: Path, query_text: str, interpolation: str = "nearest") -> np.ndarray:
        """Heat-map of <patch embedding, text embedding> over the image, at the original image size
        (`vlp/inference_engine.py:59-92`)."""
        assert not self.image_inference_engine.model.training
        assert not self.text_inference_engine.model.training
        assert isinstance(query_text, str)
        patches, (width, height) = self.image_inference_engine.get_projected_patch_embeddings(image_path)
        text = self.text_inference_engine.get_embeddings_from_prompt(query_text)
        sim = self._get_similarity_map_from_embeddings(patches, text)
        eng = self.image_inference_engine
        return self.convert_similarity_to_image_size(sim, width=width, height=height, resize_size=eng.resize_size,
                                                     crop_size=eng.crop_size, val_img_transform=eng.transform,
                                                     interpolation=interpolation)

    @staticmethod
    def _get_similarity_map_from_embeddings(projected_patch_embeddings: torch.Tensor, projected_text_embeddings: torch.Tensor,
                                            sigma: float = 1.5) -> torch.Tensor:
        """patches [h, w, D] x text [1, D] -> gaussian-smoothed similarity map [h, w] (`vlp/inference_engine.py:94-108`).
        The h*w dot products are one `cxrk_patch_similarity` launch; the smoothing is scipy on the host."""
        if projected_patch_embeddings.dim() != 3:
            raise ValueError(f"patch embeddings must be [h, w, D], got {tuple(projected_patch_embeddings.shape)}")
        h, w, d = projected_patch_embeddings.shape
        assert projected_text_embeddings.dim() == 2
        assert projected_text_embeddings.shape[0] == 1
        assert projected_text_embeddings.shape[1] == d
        sim = K.patch_similarity(projected_patch_embeddings.reshape(h * w, d), projected_text_embeddings[0])
        grid = sim.reshape(h, w).cpu().numpy()
        return torch.tensor(ndimage.gaussian_filter(grid, sigma=(sigma, sigma), order=0))

    @staticmethod
    def convert_similarity_to_image_size(similarity_map: torch.Tensor, width: int, height: int, resize_size: Optional[int],
                                         crop_size: Optional[int], val_img_transform: Optional[Callable] = None,
                                         interpolation: str = "nearest") -> np.ndarray:
        """Patch grid -> original image size, undoing the resize / centre crop the image went through
        (`vlp/inference_engine.py:110-155`): without a crop the map is stretched over the whole image; with one it is
        stretched over the crop's footprint in original pixels and the border the network never saw is NaN."""
        grid = similarity_map.reshape(1, 1, similarity_map.shape[0], similarity_map.shape[1])
        kw = dict(mode=interpolation)
        if interpolation in ("linear", "bilinear", "bicubic", "trilinear"):
            kw["align_corners"] = False
        if crop_size is None:
            return F.interpolate(grid, size=(height, width), **kw)[0, 0].numpy()
        side = crop_size if resize_size is None else int(crop_size * min(height, width) / resize_size)
        inner = F.interpolate(grid, size=(side, side), **kw)[0, 0]
        dw, dh = width - side, height - side
        pad = (math.floor(dw / 2), math.ceil(dw / 2), math.floor(dh / 2), math.ceil(dh / 2))
        return F.pad(inner, pad, value=float("nan")).numpy()

    def to(self, device: torch.device) -> None:
        self.image_inference_engine.to(device)
        self.text_inference_engine.to(device)
