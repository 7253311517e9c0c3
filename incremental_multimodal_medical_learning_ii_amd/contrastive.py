"""The north-star step: joint image + text embedding and symmetric InfoNCE, data-parallel over the GPUs of a node.

    images [B,3,224,224] --ImageModel--> I [B,128] \
                                                     > L2-normalise, all-gather, S = I_hat T_hat^T / tau, InfoNCE
    token ids [B,32]     --CXRBertModel-> T [B,128] /
    loss.backward() -> encoder backward (hand-written HIP) -> flat-gradient all-reduce (RCCL) -> fused Adam

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  Weights are replicated; the global
batch is sharded by rows.  Communication per step: all-gather of [B,256] normalised embeddings, all-gather of [B,2]
log-sum-exps, a scalar all-reduce for the reported loss, and the bucketed all-reduce of the ~133 M-parameter flat
gradient buffer.  This step is NOT in the reference: temperature is an explicit
argument (default 0.07, the usual CLIP-style value; the reference specifies none).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import functional as Fh
from . import optim as cxr_optim


class JointContrastiveTrainer:
    def __init__(self, image_model: torch.nn.Module, text_model: torch.nn.Module, lr: float = 1e-4,
                 temperature: float = 0.07, group=None, train_mlm_head: bool = False, two_streams: Optional[bool] = None,
                 optim: str = "adam"):
        self.image_model, self.text_model = image_model, text_model
        self.temperature, self.group = temperature, group
        import os
        self.two_streams = (os.environ.get("CXRK_TWO_STREAMS", "1") != "0") if two_streams is None else bool(two_streams)
        self._text_stream = None
        image_model.prepare_()
        text_model.prepare_()
        inamed = [(n, p) for n, p in image_model.named_parameters() if not n.startswith("encoder.encoder.fc.")]
        params = [p for _, p in inamed]
        tparams = []
        for n, p in text_model.named_parameters():
            if n.startswith("cls.predictions.") and not train_mlm_head:
                continue  # MLM head: no gradient on this path
            tparams.append(p)
        if optim == "adam":          # the reference's `optim.Adam(params, lr)` / `optim.SGD(params, lr)` (Trainer.py:172-178)
            self.optimizer = cxr_optim.Adam(params + tparams, lr=lr)
        elif optim == "sgd":
            self.optimizer = cxr_optim.SGD(params + tparams, lr=lr)
        else:
            raise ValueError(f"optim must be 'adam' or 'sgd', got {optim!r}")
        text_model.prepare_()
        self.world = 1
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            self.world = dist.get_world_size(group)
        self._spans = self.reduce_spans(inamed, tparams) if self.world > 1 else {}

    def reduce_spans(self, inamed=None, tparams=None) -> dict:
        """{tag: (lo, hi)} element ranges of the flat gradient buffer that become complete together during `backward()`, in the
        order they complete: "text" (the whole text encoder: its backward is a third of the step's and ends first), then the
        image encoder from the back: "head" (projector + layer4), "layer3", "layer2", "stem" (layer1 + stem).  Each range is
        all-reduced as soon as its gradients are complete (hooks on the encoders' backward, see `step`), under the rest of the
        backward; a tag whose parameters do not form one gap-free range is left to the final reduce."""
        from .image_encoder import stage_of_param
        if inamed is None:
            inamed = [(n, p) for n, p in self.image_model.named_parameters() if not n.startswith("encoder.encoder.fc.")]
        if tparams is None:
            ids = {id(p) for p in self.optimizer.params}
            tparams = [p for p in self.text_model.parameters() if id(p) in ids]
        groups = {"text": list(tparams)}
        for n, p in inamed:
            groups.setdefault(stage_of_param(n), []).append(p)
        spans = {tag: self.optimizer.grad_span(ps) for tag, ps in groups.items()}
        return {tag: sp for tag, sp in spans.items() if sp is not None}

    def forward_loss(self, images: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        """The two encoders are independent up to the loss, so the text encoder runs on a second HIP stream: the tail of one
        encoder's launch (its last, partly filled round of workgroups) overlaps the head of the other's.  Autograd replays
        each encoder's backward on the stream its forward ran on and joins them again before `backward()` returns."""
        if not (self.two_streams and images.is_cuda):
            img = self.image_model(images)
            txt = self.text_model.get_projected_text_embeddings(input_ids, attention_mask, normalize_embeddings=False)
            return Fh.infonce_loss(img, txt, self.temperature, self.group)
        if self._text_stream is None:
            self._text_stream = torch.cuda.Stream(device=images.device)
        cur = torch.cuda.current_stream(images.device)
        self._text_stream.wait_stream(cur)                     # inputs, weights and the zeroed gradients are ready
        with torch.cuda.stream(self._text_stream):
            txt = self.text_model.get_projected_text_embeddings(input_ids, attention_mask, normalize_embeddings=False)
        img = self.image_model(images)
        cur.wait_stream(self._text_stream)
        txt.record_stream(cur)
        return Fh.infonce_loss(img, txt, self.temperature, self.group)

    def step(self, images: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        """One optimisation step on this rank's shard; returns the global-batch loss (device scalar, no host sync).

        Data-parallel overlap (N > 1): the gradient all-reduce of a range of the flat buffer (`reduce_spans`) is started by the
        backward itself the moment that range is complete — a callable handed to THIS step's two encoder calls and carried by
        their autograd nodes (no module-level state): the text encoder's 0.44 GB under the image backward, then the image
        encoder's stages from the back (60 / 28 / 5 / 1 MB) under the layers in front of them; what is left (nothing, when every
        tag fired) is reduced after `backward()` returns.  Precondition, asserted: each encoder runs ONCE per step inside
        `forward_loss` — a second call of an encoder in the same graph would have its range reduced before the second
        contribution was accumulated.  A rank that raises inside `backward()` after a range was started leaves its peers inside that
        collective, as any failure of one data-parallel rank does: the job has to be torn down (torchrun / the RCCL watchdog)."""
        self.optimizer.zero_grad()
        if self.world > 1 and self._spans:
            works, fired = [], []

            def on_ready(tag: str) -> None:
                span = self._spans.get(tag)
                if span is None:
                    return
                if tag in fired:
                    raise RuntimeError(f"JointContrastiveTrainer.step: the gradients of '{tag}' were reported complete twice in one "
                                       f"backward (an encoder was called more than once in this step's graph)")
                fired.append(tag)
                works.extend(self.optimizer.all_reduce_span(span[0], span[1], self.group))

            self.image_model.grad_ready_hook = self.text_model.grad_ready_hook = on_ready
            try:
                loss = self.forward_loss(images, input_ids, attention_mask)   # the hook is captured by the two autograd nodes here
            finally:
                self.image_model.grad_ready_hook = self.text_model.grad_ready_hook = None
            loss.backward()
            self.last_overlapped = tuple(fired)
            self.optimizer.all_reduce_grads(self.group, skip=[self._spans[t] for t in fired], pending=works)
        else:
            loss = self.forward_loss(images, input_ids, attention_mask)
            loss.backward()
            if self.world > 1:
                self.optimizer.all_reduce_grads(self.group)
        self.optimizer.step()
        return loss.detach()
