"""Text-model related tools (mirror of the reference's `health_multimodal/text/__init__.py`)."""
from .inference_engine import TextInferenceEngine
from .model import CXRBertConfig, CXRBertModel, CXRBertOutput, CXRBertTokenizer
from .utils import BIOMED_VLP_CXR_BERT_SPECIALIZED, CXR_BERT_COMMIT_TAG, SyntheticTokenizer, get_cxr_bert, get_cxr_bert_inference

__all__ = ["BIOMED_VLP_CXR_BERT_SPECIALIZED", "CXR_BERT_COMMIT_TAG", "CXRBertConfig", "CXRBertTokenizer", "CXRBertModel",
           "CXRBertOutput", "TextInferenceEngine", "SyntheticTokenizer", "get_cxr_bert", "get_cxr_bert_inference"]
