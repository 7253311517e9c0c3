"""`CXRBertModel` on the cxrk HIP kernels.

Same class hierarchy, constructor, method names, outputs and parameter names as the reference's
`health_multimodal/text/model/modelling_cxrbert.py` (`BertProjectionHead` :28-49, `CXRBertModel` :52-141), so
checkpoints load with `load_state_dict`/`from_pretrained` and `TextInferenceEngine` accepts it
(`isinstance(text_model, BertForMaskedLM)`, `text/inference_engine.py:27`).  The forward (and its hand-written
backward) runs in `incremental_multimodal_medical_learning_ii_amd.text_encoder`; the HuggingFace sub-modules only
own the parameters.
"""
from dataclasses import dataclass
from typing import Any, List, Optional, Tuple, Union

import torch
from torch import Tensor as T
from torch import nn
from transformers import BertForMaskedLM
from transformers.modeling_outputs import ModelOutput

from .configuration_cxrbert import CXRBertConfig
from .... import kernels as K
from .... import text_encoder as TE

BERTTupleOutput = Tuple[T, T, T, T, T]


@dataclass
class CXRBertOutput(ModelOutput):
    last_hidden_state: torch.FloatTensor = None
    logits: Optional[torch.FloatTensor] = None
    cls_projected_embedding: Optional[torch.FloatTensor] = None
    hidden_states: Optional[Tuple[torch.FloatTensor]] = None
    attentions: Optional[Tuple[torch.FloatTensor]] = None


class BertProjectionHead(nn.Module):
    """Projection head for the BERT CLS token: Linear -> GELU -> LayerNorm(eps 1e-12) -> Linear
    (reference `modelling_cxrbert.py:28-49`).  Parameter container; executed by `text_encoder`."""

    def __init__(self, config: CXRBertConfig) -> None:
        super().__init__()
        self.dense_to_hidden = nn.Linear(config.hidden_size, config.projection_size)
        self.transform_act_fn = nn.functional.gelu
        self.LayerNorm = nn.LayerNorm(config.projection_size, eps=1e-12)
        self.dense_to_output = nn.Linear(config.projection_size, config.projection_size)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        h1_pre = torch.empty(hidden_states.shape[0], self.dense_to_hidden.out_features, dtype=torch.float32,
                             device=hidden_states.device)
        with torch.no_grad():
            h1 = K.linear_fwd(hidden_states, self.dense_to_hidden.weight, self.dense_to_hidden.bias, act=K.ACT_GELU,
                              preact_out=h1_pre)
            h2, _, _ = K.residual_ln_fwd(h1, None, self.LayerNorm.weight, self.LayerNorm.bias, 1e-12, save=False)
            return K.linear_fwd(h2, self.dense_to_output.weight, self.dense_to_output.bias)


class CXRBertModel(BertForMaskedLM):
    """CXR-BERT (Boecking et al. 2022) = HuggingFace BertForMaskedLM + a CLS projection head
    (reference `modelling_cxrbert.py:52-141`)."""

    config_class = CXRBertConfig  # type: ignore

    def __init__(self, config: CXRBertConfig):
        super().__init__(config)
        self.cls_projection_head = BertProjectionHead(config)
        if hasattr(self, "post_init"):
            self.post_init()
        else:  # transformers 4.17 spelling used by the reference (:68)
            self.init_weights()
        self._hot: Optional[List[nn.Parameter]] = None
        # optional callable(tag): handed to every grad-enabled encode call made while it is set and called by THAT call's backward
        # when its parameter gradients are complete (text_encoder.encode); set and cleared by the data-parallel trainer per step
        self.grad_ready_hook = None

    # ------------------------------------------------------------------ parameter plumbing
    def _hot_params(self) -> List[nn.Parameter]:
        if self._hot is None:
            named = dict(self.named_parameters())
            self._hot = [named[n] for n in TE.param_names(self.config.num_hidden_layers)]
        return self._hot

    def prepare_(self) -> "CXRBertModel":
        """Fuse every layer's query/key/value parameters into one [3H,H] buffer (idempotent; values unchanged).
        Needed again after `.to(device)` because that re-allocates parameter storage."""
        for layer in self.bert.encoder.layer:
            att = layer.attention.self
            TE.fuse_qkv_(att.query.weight, att.key.weight, att.value.weight)
            TE.fuse_qkv_(att.query.bias, att.key.bias, att.value.bias)
        return self

    def _check_mode(self) -> None:
        """Dropout is not implemented; the reference only ever runs the text model in eval mode
        (text/inference_engine.py:63 asserts it).  Refuse rather than silently skip the dropout of a training-mode model."""
        cfg = self.config
        if self.training and (getattr(cfg, "hidden_dropout_prob", 0.0) > 0 or getattr(cfg, "attention_probs_dropout_prob", 0.0) > 0):
            raise NotImplementedError("CXRBertModel is in training mode with dropout > 0, which the HIP path does not implement; "
                                      "call .eval() (parameters still receive gradients) or set the dropout probabilities to 0")

    def _encode(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], cls_only: bool = False, want_last: bool = True):
        if not input_ids.is_cuda:
            raise RuntimeError("CXRBertModel runs on the MI355X only: move the model and inputs to 'cuda' "
                               "(there is no CPU fallback; the CPU oracle lives in oracle/ and is test-only)")
        self.prepare_()
        cfg = self.config
        self._check_mode()
        if getattr(cfg, "hidden_act", "gelu") != "gelu":
            raise NotImplementedError(f"hidden_act={cfg.hidden_act!r}: only erf-GELU (CXR-BERT) is implemented")
        return TE.encode(self._hot_params(), input_ids, attention_mask, cfg.num_hidden_layers,
                         cfg.num_attention_heads, cfg.layer_norm_eps, cls_only, want_last,
                         on_grads_ready=self.grad_ready_hook if torch.is_grad_enabled() else None)

    @torch.no_grad()
    def _mlm_logits(self, last_hidden: torch.Tensor) -> torch.Tensor:
        """HF `BertOnlyMLMHead`: transform (dense, gelu, LayerNorm) + decoder tied to the word embeddings."""
        tr = self.cls.predictions.transform
        N, L, H = last_hidden.shape
        h = K.linear_fwd(last_hidden.reshape(N * L, H), tr.dense.weight, tr.dense.bias, act=K.ACT_GELU)
        h, _, _ = K.residual_ln_fwd(h, None, tr.LayerNorm.weight, tr.LayerNorm.bias, self.config.layer_norm_eps,
                                    save=False)
        dec = self.cls.predictions.decoder
        bias = self.cls.predictions.bias if getattr(dec, "bias", None) is None else dec.bias
        return K.linear_fwd(h, dec.weight, bias).view(N, L, -1)

    # ------------------------------------------------------------------ reference API
    def forward(
        self,
        input_ids: torch.Tensor,
        attention_mask: torch.Tensor,
        token_type_ids: Optional[torch.Tensor] = None,
        position_ids: Optional[torch.Tensor] = None,
        head_mask: Optional[torch.Tensor] = None,
        inputs_embeds: Optional[torch.Tensor] = None,
        output_attentions: Optional[bool] = None,
        output_hidden_states: Optional[bool] = None,
        output_cls_projected_embedding: Optional[bool] = None,
        return_dict: Optional[bool] = None,
        output_mlm_logits: bool = True,
        **kwargs: Any
    ) -> Union[BERTTupleOutput, CXRBertOutput]:
        for name, val in (("token_type_ids", token_type_ids), ("position_ids", position_ids), ("head_mask", head_mask),
                          ("inputs_embeds", inputs_embeds)):
            if val is not None:
                raise NotImplementedError(f"{name} is not supported by the HIP path (the reference never passes it: "
                                          f"text/inference_engine.py:65-68)")
        if output_attentions:
            raise NotImplementedError("output_attentions is not supported by the HIP path")
        return_dict = return_dict if return_dict is not None else getattr(self.config, "use_return_dict", True)
        proj, last_hidden_state = self._encode(input_ids, attention_mask)
        cls_projected_embedding = proj if output_cls_projected_embedding else None
        logits = self._mlm_logits(last_hidden_state.detach()) if output_mlm_logits else None
        hidden_states = (last_hidden_state,) if output_hidden_states else None  # only the last layer is materialised
        if return_dict:
            return CXRBertOutput(last_hidden_state=last_hidden_state, logits=logits,
                                 cls_projected_embedding=cls_projected_embedding, hidden_states=hidden_states,
                                 attentions=None)
        return (last_hidden_state, logits, cls_projected_embedding, hidden_states, None)

    def get_projected_text_embeddings(self, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                                      normalize_embeddings: bool = True) -> torch.Tensor:
        """Projected CLS embeddings [batch, projection_size], optionally L2-normalised
        (reference `modelling_cxrbert.py:117-141`)."""
        # Only hidden_states[-1][:, 0, :] feeds the projection head (:98-99): the last layer's row-wise part runs on the CLS
        # rows alone (text_encoder._forward, cls_only).  Same values as forward(...).cls_projected_embedding.
        cls_projected_embedding, _ = self._encode(input_ids, attention_mask, cls_only=True, want_last=False)
        if normalize_embeddings:
            from ....functional import l2_normalize
            return l2_normalize(cls_projected_embedding)
        return cls_projected_embedding
