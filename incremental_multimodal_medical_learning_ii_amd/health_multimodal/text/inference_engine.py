"""`TextInferenceEngine` — same surface as the reference's `health_multimodal/text/inference_engine.py:14-119`.
Tokenised prompts are cached (the reference re-tokenises the same 40 constant prompts 10x per step)."""
from typing import Any, Dict, List, Tuple, Union

import torch
from transformers import BertForMaskedLM

from .data.io import TextInput


class TextInferenceEngine(TextInput):
    """Sentence embeddings, pairwise similarities and MLM predictions from a CXR-BERT model.

    :param tokenizer: A BertTokenizer-like object.
    :param text_model: A `BertForMaskedLM` (here: the HIP-backed `CXRBertModel`).
    """

    def __init__(self, tokenizer: Any, text_model: BertForMaskedLM) -> None:
        super().__init__(tokenizer=tokenizer)

        assert isinstance(text_model, BertForMaskedLM), f"Expected a BertForMaskedLM, got {type(text_model)}"

        self.model = text_model
        self.max_allowed_input_length = self.model.config.max_position_embeddings
        self.to = self.model.to
        self._tok_cache: Dict[Tuple[str, ...], Any] = {}

    def is_in_eval(self) -> bool:
        """Returns True if the model is in eval mode."""
        return not self.model.training

    def tokenize_input_prompts(self, prompts: Union[str, List[str]], verbose: bool = True) -> Any:
        key = (prompts,) if isinstance(prompts, str) else tuple(prompts)
        device = next(self.model.parameters()).device
        hit = self._tok_cache.get(key)
        if hit is not None and hit.input_ids.device == device:
            return hit
        tokenizer_output = super().tokenize_input_prompts(prompts, verbose=verbose)
        tokenizer_output.input_ids = tokenizer_output.input_ids.to(device)
        tokenizer_output.attention_mask = tokenizer_output.attention_mask.to(device)

        max_length = tokenizer_output.input_ids.shape[1]
        if tokenizer_output.input_ids.shape[1] > self.max_allowed_input_length:
            raise ValueError(f"The sequence length of the input ({max_length}) is "
                             f"longer than the maximum allowed sequence length ({self.max_allowed_input_length}).")
        if len(self._tok_cache) < 4096:
            self._tok_cache[key] = tokenizer_output
        return tokenizer_output

    @torch.no_grad()
    def get_embeddings_from_prompt(self, prompts: Union[str, List[str]], normalize: bool = True,
                                   verbose: bool = True) -> torch.Tensor:
        """L2-normalised (optionally) projected embeddings [batch, embedding_size] for text prompt(s)."""
        assert self.is_in_eval()
        tokenizer_output = self.tokenize_input_prompts(prompts=prompts, verbose=verbose)
        txt_emb = self.model.get_projected_text_embeddings(  # type: ignore
            input_ids=tokenizer_output.input_ids,
            attention_mask=tokenizer_output.attention_mask,
            normalize_embeddings=normalize)
        return txt_emb

    @torch.no_grad()
    def get_pairwise_similarities(self, prompt_set_1: Union[str, List[str]],
                                  prompt_set_2: Union[str, List[str]]) -> torch.Tensor:
        """Cosine similarity between the i-th prompts of the two sets."""
        from ... import kernels as K
        emb_1 = self.get_embeddings_from_prompt(prompts=prompt_set_1, verbose=False)
        emb_2 = self.get_embeddings_from_prompt(prompts=prompt_set_2, verbose=False)
        n = emb_1.shape[0]
        full = torch.empty(n, emb_2.shape[0], dtype=torch.float32, device=emb_1.device)
        K.gemm(emb_1, emb_2, full, n, emb_2.shape[0], emb_1.shape[1], False, True)
        return torch.diag(full).detach()

    @torch.no_grad()
    def predict_masked_tokens(self, prompts: Union[str, List[str]]) -> List[List[str]]:
        """Top-1 token candidates at the [MASK] positions (needs an MLM head)."""
        assert self.is_in_eval()
        tokenized_prompts = self.tokenize_input_prompts(prompts)
        text_model_output = self.model.forward(input_ids=tokenized_prompts.input_ids,
                                               attention_mask=tokenized_prompts.attention_mask)
        logits = text_model_output.logits.detach()
        predicted_token_ids = torch.argmax(logits, dim=-1)  # Batch x Seq
        batch_size = predicted_token_ids.shape[0]
        mask_token_id = self.tokenizer.mask_token_id
        mlm_mask = tokenized_prompts.input_ids == mask_token_id  # Batch x Seq
        output = list()
        for b in range(batch_size):
            _ids = predicted_token_ids[b, mlm_mask[b]].cpu().tolist()
            _tokens = self.tokenizer.convert_ids_to_tokens(_ids)
            output.append(_tokens)
        return output
