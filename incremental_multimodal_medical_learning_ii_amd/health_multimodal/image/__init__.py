"""Image-model related tools (mirror of the reference's `health_multimodal/image/__init__.py`)."""
from .inference_engine import ImageInferenceEngine
from .model import BIOVIL_IMAGE_WEIGHTS_NAME, ImageEncoder, ImageModel, ResnetType, get_biovil_resnet
from .utils import get_biovil_resnet_inference

__all__ = ["BIOVIL_IMAGE_WEIGHTS_NAME", "ImageEncoder", "ImageInferenceEngine", "ImageModel", "ResnetType",
           "get_biovil_resnet", "get_biovil_resnet_inference"]
