from .model import (BIOMED_VLP_CXR_BERT_SPECIALIZED, BIOVIL_IMAGE_WEIGHTS_NAME, CXR_BERT_COMMIT_TAG, ImageEncoder, ImageModel,
                    ImageModelOutput, JOINT_FEATURE_SIZE, MODEL_TYPE, ResnetType, get_biovil_resnet)
from .modules import MLP
from .resnet import ResNetHIML, resnet50

__all__ = ["BIOMED_VLP_CXR_BERT_SPECIALIZED", "CXR_BERT_COMMIT_TAG", "BIOVIL_IMAGE_WEIGHTS_NAME", "ImageModel", "ImageEncoder",
           "ImageModelOutput", "ResnetType", "get_biovil_resnet", "MLP", "ResNetHIML", "resnet50", "MODEL_TYPE",
           "JOINT_FEATURE_SIZE"]
