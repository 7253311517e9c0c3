"""Factories — same names as the reference's `health_multimodal/text/utils.py:15-35`.

The reference fetches tokenizer and weights from the Hugging Face Hub ("microsoft/BiomedVLP-CXR-BERT-specialized",
revision v1.1).  That stays the default, but both pieces can be injected (local directory, pre-built objects,
or the offline synthetic pair) because the GPU box has no network."""
import os
import re
import zlib
from types import SimpleNamespace
from typing import Any, List, Optional, Tuple

import torch

from .inference_engine import TextInferenceEngine
from .model import CXRBertConfig, CXRBertModel, CXRBertTokenizer

BIOMED_VLP_CXR_BERT_SPECIALIZED = "microsoft/BiomedVLP-CXR-BERT-specialized"
CXR_BERT_COMMIT_TAG = "v1.1"


class SyntheticTokenizer:
    """Offline stand-in with the BertTokenizer methods the engine uses.  Words are hashed into the vocabulary
    (stable across processes); ids 0..4 are [PAD] [UNK] [CLS] [SEP] [MASK]."""

    pad_token, unk_token, cls_token, sep_token, mask_token = "[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"
    pad_token_id, unk_token_id, cls_token_id, sep_token_id, mask_token_id = 0, 1, 2, 3, 4

    def __init__(self, vocab_size: int = 30522):
        self.vocab_size = vocab_size
        self._rev = {0: "[PAD]", 1: "[UNK]", 2: "[CLS]", 3: "[SEP]", 4: "[MASK]"}

    @property
    def all_special_tokens(self) -> List[str]:
        return [self.unk_token, self.sep_token, self.pad_token, self.cls_token, self.mask_token]

    def _word_id(self, w: str) -> int:
        if w == self.mask_token:
            return self.mask_token_id
        i = 5 + zlib.crc32(w.encode("utf-8")) % (self.vocab_size - 5)
        self._rev.setdefault(i, w)
        return i

    def tokenize(self, text: str) -> List[str]:
        return re.findall(r"\[MASK\]|\w+|[^\w\s]", text.lower().replace("[mask]", "[MASK]"))

    def batch_encode_plus(self, batch_text_or_text_pairs, add_special_tokens=True, padding='longest',
                          return_tensors='pt', **_: Any):
        rows = []
        for t in batch_text_or_text_pairs:
            ids = [self._word_id(w) for w in self.tokenize(t)]
            rows.append([self.cls_token_id] + ids + [self.sep_token_id] if add_special_tokens else ids)
        n = max(len(r) for r in rows)
        ids = torch.full((len(rows), n), self.pad_token_id, dtype=torch.int64)
        mask = torch.zeros(len(rows), n, dtype=torch.int64)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r, dtype=torch.int64)
            mask[i, :len(r)] = 1
        return SimpleNamespace(input_ids=ids, attention_mask=mask, token_type_ids=torch.zeros_like(ids))

    def convert_ids_to_tokens(self, ids: List[int]) -> List[str]:
        return [self._rev.get(int(i), f"[id{int(i)}]") for i in ids]


def get_cxr_bert(pretrained: Optional[str] = None) -> Tuple[Any, CXRBertModel]:
    """Load tokenizer + model.  `pretrained` = hub name (default, needs network), a local directory, or
    "synthetic" for name-keyed synthetic weights and the hashing tokenizer (offline)."""
    if pretrained == "synthetic" or os.environ.get("CXRK_SYNTHETIC_WEIGHTS") == "1":
        from ...synthetic import fill_module_
        model = CXRBertModel(CXRBertConfig())
        fill_module_(model)
        model.eval()
        return SyntheticTokenizer(model.config.vocab_size), model
    model_name = pretrained or BIOMED_VLP_CXR_BERT_SPECIALIZED
    kw = {} if pretrained else {"revision": CXR_BERT_COMMIT_TAG}
    tokenizer = CXRBertTokenizer.from_pretrained(model_name, **kw)
    text_model = CXRBertModel.from_pretrained(model_name, **kw)
    return tokenizer, text_model


def get_cxr_bert_inference(pretrained: Optional[str] = None, device: Optional[str] = None) -> TextInferenceEngine:
    """Create a :class:`TextInferenceEngine` for the CXR-BERT model (reference `text/utils.py:25-35`)."""
    tokenizer, text_model = get_cxr_bert(pretrained)
    text_model.eval()
    if device is not None:
        text_model.to(device)
    text_inference = TextInferenceEngine(tokenizer=tokenizer, text_model=text_model)
    if text_inference.is_in_eval():
        print("*** Bert is in eval mode ***")
    return text_inference
