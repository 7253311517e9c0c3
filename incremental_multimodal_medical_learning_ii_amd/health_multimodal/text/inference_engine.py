"""`TextInferenceEngine`: prompt strings -> projected CXR-BERT embeddings on the HIP path.

Mirrors the surface of the reference's engine (`health_multimodal/text/inference_engine.py:14-119`: constructor,
`is_in_eval`, `tokenize_input_prompts`, `get_embeddings_from_prompt`, `get_pairwise_similarities`,
`predict_masked_tokens`, `.to`) with two differences a drop-in user can rely on: tokenised prompts are memoised per
device (the reference's training loop re-tokenises the same 40 constant prompts ten times per step),
and the similarity of paired prompts is one logits GEMM on the MFMA core instead of a host-side matmul.
"""
from typing import Any, Dict, List, Tuple, Union

import torch
from transformers import BertForMaskedLM

from .data.io import TextInput, TypePrompts

_CACHE_LIMIT = 4096


class TextInferenceEngine(TextInput):
    def __init__(self, tokenizer: Any, text_model: BertForMaskedLM) -> None:
        if not isinstance(text_model, BertForMaskedLM):
            raise AssertionError(f"Expected a BertForMaskedLM, got {type(text_model)}")
        super().__init__(tokenizer=tokenizer)
        self.model = text_model
        self.max_allowed_input_length = int(text_model.config.max_position_embeddings)
        self.to = text_model.to                       # engine.to(device) moves the model, as in the reference
        self._tok_cache: Dict[Tuple[str, ...], Any] = {}

    # ------------------------------------------------------------------ state
    def is_in_eval(self) -> bool:
        return not self.model.training

    def _device(self) -> torch.device:
        return next(self.model.parameters()).device

    # ------------------------------------------------------------------ tokenisation (memoised)
    def tokenize_input_prompts(self, prompts: TypePrompts, verbose: bool = True) -> Any:
        key = (prompts,) if isinstance(prompts, str) else tuple(prompts)
        device = self._device()
        cached = self._tok_cache.get(key)
        if cached is not None and cached.input_ids.device == device:
            return cached
        encoded = TextInput.tokenize_input_prompts(self, prompts, verbose=verbose)
        n_tokens = int(encoded.input_ids.shape[1])
        if n_tokens > self.max_allowed_input_length:
            raise ValueError(f"The sequence length of the input ({n_tokens}) is longer than the maximum allowed sequence "
                             f"length ({self.max_allowed_input_length}).")
        encoded.input_ids = encoded.input_ids.to(device)
        encoded.attention_mask = encoded.attention_mask.to(device)
        if len(self._tok_cache) < _CACHE_LIMIT:
            self._tok_cache[key] = encoded
        return encoded

    # ------------------------------------------------------------------ embeddings
    @torch.no_grad()
    def get_embeddings_from_prompt(self, prompts: TypePrompts, normalize: bool = True, verbose: bool = True) -> torch.Tensor:
        """[n_prompts, projection_size] projected CLS embeddings (L2-normalised when `normalize`).  Eval mode only."""
        assert self.is_in_eval()
        enc = self.tokenize_input_prompts(prompts, verbose=verbose)
        return self.model.get_projected_text_embeddings(input_ids=enc.input_ids, attention_mask=enc.attention_mask,
                                                        normalize_embeddings=normalize)

    @torch.no_grad()
    def get_pairwise_similarities(self, prompt_set_1: TypePrompts, prompt_set_2: TypePrompts) -> torch.Tensor:
        """cos(prompt_set_1[i], prompt_set_2[i]) for every i: the diagonal of the normalised-embedding logits."""
        from ... import kernels as K
        left = self.get_embeddings_from_prompt(prompt_set_1, verbose=False)
        right = self.get_embeddings_from_prompt(prompt_set_2, verbose=False)
        logits = torch.empty(left.shape[0], right.shape[0], dtype=torch.float32, device=left.device)
        K.gemm(left, right, logits, left.shape[0], right.shape[0], left.shape[1], False, True)
        return logits.diagonal().clone()

    # ------------------------------------------------------------------ masked-token prediction
    @torch.no_grad()
    def predict_masked_tokens(self, prompts: TypePrompts) -> List[List[str]]:
        """For each prompt, the arg-max vocabulary token at each of its [MASK] positions."""
        assert self.is_in_eval()
        enc = self.tokenize_input_prompts(prompts)
        best = self.model.forward(input_ids=enc.input_ids, attention_mask=enc.attention_mask).logits.argmax(dim=-1)
        at_mask = enc.input_ids == self.tokenizer.mask_token_id
        return [self.tokenizer.convert_ids_to_tokens(row[sel].cpu().tolist()) for row, sel in zip(best, at_mask)]
