"""Visual-language processing tools (mirror of the reference's `health_multimodal/vlp/__init__.py`)."""
from .inference_engine import ImageTextInferenceEngine

__all__ = ["ImageTextInferenceEngine"]
