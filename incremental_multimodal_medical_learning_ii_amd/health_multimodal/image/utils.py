"""Factory — same name as the reference's `image/utils.py:15-27` (which, as shipped, calls `get_biovil_resnet()`
without its now-required argument: SURVEY.md F10)."""
from pathlib import Path
from typing import Optional, Union

from .data.transforms import create_chest_xray_transform_for_inference
from .inference_engine import ImageInferenceEngine
from .model import get_biovil_resnet

TRANSFORM_RESIZE = 512
TRANSFORM_CENTER_CROP_SIZE = 480


def get_biovil_resnet_inference(pretrained: Optional[Union[str, Path]] = None) -> ImageInferenceEngine:
    """Create an :class:`ImageInferenceEngine`; `pretrained` = path of `biovil_image_resnet50_proj_size_128.pt`."""
    image_model = get_biovil_resnet(pretrained)
    transform = create_chest_xray_transform_for_inference(resize=TRANSFORM_RESIZE,
                                                          center_crop_size=TRANSFORM_CENTER_CROP_SIZE)
    return ImageInferenceEngine(image_model=image_model, transform=transform)
