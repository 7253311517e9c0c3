"""`ImageTextInferenceEngine` — same surface as the reference's `vlp/inference_engine.py:21-57,157-160`.
The similarity-map / heat-map helpers (`:59-156`) are visualisation and out of the hot path."""
from pathlib import Path
from typing import List, Union

import torch

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

    def to(self, device: torch.device) -> None:
        self.image_inference_engine.to(device)
        self.text_inference_engine.to(device)
