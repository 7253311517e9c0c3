from .io import load_image
from .transforms import create_chest_xray_transform_for_inference, infer_resize_params

__all__ = ["load_image", "create_chest_xray_transform_for_inference", "infer_resize_params"]
