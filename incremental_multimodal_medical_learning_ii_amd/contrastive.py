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
                 temperature: float = 0.07, group=None, train_mlm_head: bool = False, two_streams: Optional[bool] = None):
        self.image_model, self.text_model = image_model, text_model
        self.temperature, self.group = temperature, group
        import os
        self.two_streams = (os.environ.get("CXRK_TWO_STREAMS", "1") != "0") if two_streams is None else bool(two_streams)
        self._text_stream = None
        image_model.prepare_()
        text_model.prepare_()
        params = [p for n, p in image_model.named_parameters() if not n.startswith("encoder.encoder.fc.")]
        tparams = []
        for n, p in text_model.named_parameters():
            if n.startswith("cls.predictions.") and not train_mlm_head:
                continue  # MLM head: no gradient on this path
            tparams.append(p)
        self.optimizer = cxr_optim.Adam(params + tparams, lr=lr)
        text_model.prepare_()
        self.world = 1
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            self.world = dist.get_world_size(group)
        # Data-parallel overlap: the text encoder's backward (~1/3 of the step's backward) finishes long before the image
        # encoder's; its 0.44 GB of gradients are one gap-free range of the flat buffer and are all-reduced as soon as they are
        # complete (hook at the end of the text backward, on the text stream), under the image backward.  Only the image
        # encoder's 0.1 GB remains to be reduced after backward() returns.
        self._text_span = self.optimizer.grad_span(tparams) if self.world > 1 else None
        self._early = None

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
        """One optimisation step on this rank's shard; returns the global-batch loss (device scalar, no host sync)."""
        self.optimizer.zero_grad()
        loss = self.forward_loss(images, input_ids, attention_mask)
        if self.world > 1 and self._text_span is not None:
            from . import text_encoder as TE
            self._early = None

            def start_text_reduce():
                self._early = self.optimizer.all_reduce_span(self._text_span[0], self._text_span[1], self.group)

            TE.after_backward = start_text_reduce
            try:
                loss.backward()
            finally:
                TE.after_backward = None
            early = self._early
            self.optimizer.all_reduce_grads(self.group, skip=self._text_span if early is not None else None, pending=early)
        else:
            loss.backward()
            if self.world > 1:
                self.optimizer.all_reduce_grads(self.group)
        self.optimizer.step()
        return loss.detach()
