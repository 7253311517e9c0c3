"""`ImageModel` / `ImageEncoder` / `get_biovil_resnet` — same surface as the reference's
`health_multimodal/image/model/model.py` (factory :61-70, `ImageModel` :88-173, `ImageEncoder` :176-227), executed by
the cxrk HIP kernels through `image_encoder.ImageEncodeFn`."""
from __future__ import annotations

import enum
from dataclasses import dataclass
from pathlib import Path
from typing import Any, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from .modules import MLP
from .resnet import resnet18, resnet50
from .... import image_encoder as IE
from .... import kernels as K

TypeImageEncoder = Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]
MODEL_TYPE = "resnet50"
JOINT_FEATURE_SIZE = 128

BIOMED_VLP_CXR_BERT_SPECIALIZED = "microsoft/BiomedVLP-CXR-BERT-specialized"
REPO_URL = f"https://huggingface.co/{BIOMED_VLP_CXR_BERT_SPECIALIZED}"
CXR_BERT_COMMIT_TAG = "v1.1"
BIOVIL_IMAGE_WEIGHTS_NAME = "biovil_image_resnet50_proj_size_128.pt"
BIOVIL_IMAGE_WEIGHTS_URL = f"{REPO_URL}/resolve/{CXR_BERT_COMMIT_TAG}/{BIOVIL_IMAGE_WEIGHTS_NAME}"
BIOVIL_IMAGE_WEIGHTS_MD5 = "02ce6ee460f72efd599295f440dbb453"


def get_biovil_resnet(pretrained: Optional[Union[str, Path]] = None) -> "ImageModel":
    """Instantiate the BioViL image model; `pretrained` is a local checkpoint path (reference :61-70) or None."""
    return ImageModel(img_model_type=MODEL_TYPE, joint_feature_size=JOINT_FEATURE_SIZE,
                      pretrained_model_path=pretrained)


@enum.unique
class ResnetType(str, enum.Enum):
    RESNET18 = "resnet18"
    RESNET50 = "resnet50"


@dataclass
class ImageModelOutput():
    img_embedding: torch.Tensor
    patch_embedding: torch.Tensor
    projected_global_embedding: torch.Tensor
    class_logits: torch.Tensor
    projected_patch_embeddings: torch.Tensor


class ImageEncoder(nn.Module):
    """Trunk (`model.py:176-227`): owns `self.encoder` (ResNetHIML)."""

    def __init__(self, img_model_type: str):
        super().__init__()
        self.img_model_type = img_model_type
        self.encoder = self._create_encoder()

    def _create_encoder(self, **kwargs: Any) -> nn.Module:
        supported = ResnetType.RESNET18, ResnetType.RESNET50
        if self.img_model_type not in supported:
            raise NotImplementedError(f"Image model type \"{self.img_model_type}\" must be in {supported}")
        encoder_class = resnet18 if self.img_model_type == ResnetType.RESNET18 else resnet50
        return encoder_class(pretrained=False, **kwargs)

    def forward(self, x: torch.Tensor, return_patch_embeddings: bool = False) -> TypeImageEncoder:
        raise RuntimeError("call ImageModel.forward: trunk + projector run as one fused autograd function")


class ImageModel(nn.Module):
    """Image encoder module (`model.py:88-173`): `forward(x[B,3,H,W]) -> projected global embedding [B,128]`."""

    def __init__(self, img_model_type: str, joint_feature_size: int, freeze_encoder: bool = False,
                 pretrained_model_path: Optional[Union[str, Path]] = None, **downstream_classifier_kwargs: Any):
        super().__init__()
        if downstream_classifier_kwargs:
            raise NotImplementedError("downstream classifier heads are outside the hot path (model.py:156-159)")
        self.encoder = ImageEncoder(img_model_type)
        self.feature_size = 2048  # the reference probes this with a [1,3,32,32] forward (:101,231-247)
        self.projector = MLP(input_dim=self.feature_size, output_dim=joint_feature_size,
                             hidden_dim=joint_feature_size, use_1x1_convs=True)
        self.downstream_classifier_kwargs = downstream_classifier_kwargs
        self.classifier = None
        self.freeze_encoder = freeze_encoder
        self.train()
        self._specs, self._blocks = IE.resnet50_specs("encoder.encoder.", joint_feature_size)
        self._hot: Optional[Tuple[List[nn.Parameter], List[torch.Tensor]]] = None
        self._bn: Optional[List[nn.Module]] = None
        # optional callable(stage): handed to every grad-enabled forward made while it is set and called by THAT forward's backward
        # as the gradients of "head" (projector + layer4), "layer3", "layer2", "stem" (layer1 + stem) become complete
        # (image_encoder._backward); set and cleared by the data-parallel trainer per step
        self.grad_ready_hook = None
        if pretrained_model_path is not None:
            if not isinstance(pretrained_model_path, (str, Path)):
                raise TypeError(f"Expected a string or Path, got {type(pretrained_model_path)}")
            state_dict = torch.load(pretrained_model_path, map_location="cpu", weights_only=True)
            self.load_state_dict(state_dict)

    def train(self, mode: bool = True, my_freeze: bool = False) -> Any:
        """Switch between training and evaluation modes (`model.py:131-139`).  In training mode the BatchNorm layers normalise with
        batch statistics and update their running statistics (csrc/bn_train.hip); `.eval()` / `my_freeze=True` — what every
        reference call site uses — run them on the running statistics folded into the filters (the fast path `bench.py` measures)."""
        super().train(mode=mode)
        if my_freeze:
            print("freezing resnet encoder and projector")
            self.encoder.train(mode=False)
            self.projector.train(mode=False)
        return self

    # ------------------------------------------------------------------ parameter plumbing
    def prepare_(self) -> "ImageModel":
        """Put every conv filter in channels_last memory ([Ko][R][S][C]); idempotent, values unchanged."""
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Conv2d) and not m.weight.data.is_contiguous(memory_format=torch.channels_last):
                    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        return self

    def _tensors(self):
        if self._hot is None:
            named = dict(self.named_parameters())
            bufs = dict(self.named_buffers())
            self._hot = ([named[n] for n in IE.param_names(self._specs)], [bufs[n] for n in IE.buffer_names(self._specs)])
        return self._hot

    def _bn_mode(self) -> Optional[float]:
        """None when every BatchNorm layer is in eval mode (running statistics: the only mode the reference ever runs the encoder in,
        chexpert-get-embedding.py:41-42); their common momentum when every one is in training mode (batch statistics + running-stat
        updates: what `self.train()` at the end of the reference's constructor selects).  Mixed states and `momentum=None`
        (cumulative averaging) are refused."""
        if self._bn is None:
            self._bn = [m for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        flags = {m.training for m in self._bn}
        if flags == {False}:
            return None
        if flags != {True}:
            raise NotImplementedError("ImageModel: some BatchNorm layers are in training mode and some in eval mode; the HIP path runs "
                                      "them all on batch statistics (.train()) or all on running statistics (.eval() / .train(my_freeze=True))")
        moms = {m.momentum for m in self._bn}
        if len(moms) != 1 or None in moms:
            raise NotImplementedError(f"ImageModel: train-mode BatchNorm needs one float momentum for all layers, got {moms}")
        return float(moms.pop())

    def _require_eval(self, what: str) -> None:
        if self._bn_mode() is not None:
            raise NotImplementedError(f"ImageModel.{what} runs on the running statistics: call .eval() first")

    def _run(self, x: torch.Tensor, want_patch: bool):
        if not x.is_cuda:
            raise RuntimeError("ImageModel runs on the MI355X only: move the model and inputs to 'cuda' "
                               "(there is no CPU fallback; the CPU oracle lives in oracle/ and is test-only)")
        if x.dtype != torch.float32:
            raise ValueError(f"expected fp32 images, got {x.dtype}")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"ImageModel expects [B,3,H,W] input (ExpandChannels, transforms.py:12-38), got {tuple(x.shape)}")
        momentum = self._bn_mode()
        self.prepare_()
        params, bufs = self._tensors()
        hook = self.grad_ready_hook if torch.is_grad_enabled() else None
        meta = (self._specs, self._blocks, len(params), want_patch, hook, momentum)
        with torch.set_grad_enabled(torch.is_grad_enabled() and not self.freeze_encoder):
            emb, patch = IE.ImageEncodeFn.apply(x, meta, *params, *bufs)
        if momentum is not None:       # the kernels updated running_mean / running_var in place; the counter is host bookkeeping
            with torch.no_grad():
                for m in self._bn:
                    m.num_batches_tracked += 1
        return emb, patch

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Projected global embedding [B, joint_feature_size] (the reference's edited forward, `model.py:141-154`)."""
        emb, _ = self._run(x, want_patch=False)
        return emb

    @torch.no_grad()
    def project_patch_embeddings(self, patch_embeddings: torch.Tensor) -> torch.Tensor:
        """The projector alone (reference `modules.MLP`, modules.py:29-47): trunk patch embeddings [B,2048,h,w] ->
        projected patch embeddings [B,joint,h,w]."""
        self._require_eval("project_patch_embeddings")
        params, bufs = self._tensors()
        return IE.project_patches(self._specs, params, bufs, patch_embeddings)

    @torch.no_grad()
    def calibrate_batchnorm_(self, x: torch.Tensor) -> "ImageModel":
        """Replace every BatchNorm's running statistics by the statistics of its input over the batch `x` (one train-mode pass
        with momentum 1, on the HIP kernels).  Not part of the reference's surface: it gives synthetic weights the matched
        statistics a trained checkpoint has (bench.py)."""
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"calibrate_batchnorm_: expected fp32 [B,3,H,W] images on the GPU, got {x.dtype} {tuple(x.shape)} on {x.device}")
        self.prepare_()
        params, bufs = self._tensors()
        IE.calibrate_batchnorm_(self._specs, self._blocks, params, bufs, x)
        return self

    @torch.no_grad()
    def forward_stages(self, x: torch.Tensor):
        """Diagnostic: [max-pooled stem, layer1..layer4] outputs (fp32 NCHW) from the kernels `forward` runs."""
        self._require_eval("forward_stages")
        self.prepare_()
        params, bufs = self._tensors()
        return IE.forward_stages(self._specs, self._blocks, params, bufs, x)

    @torch.no_grad()
    def get_patchwise_projected_embeddings(self, input_img: torch.Tensor, normalize: bool) -> torch.Tensor:
        """Patch-wise projected embeddings [batch, n_patches_h, n_patches_w, feature_size] (`model.py:161-173`)."""
        assert not self.training, "This function is only implemented for evaluation mode"
        _, patch = self._run(input_img, want_patch=True)  # already B H W D
        if normalize:
            n, h, w, d = patch.shape
            xhat, _ = K.l2norm_fwd(patch.reshape(n * h * w, d))
            patch = xhat.view(n, h, w, d)
        return patch


@torch.no_grad()
def get_encoder_output_dim(module: torch.nn.Module) -> int:
    """Output feature dimension of the trunk (`model.py:231-247`); static for ResNet-50."""
    return 2048
