from .configuration_cxrbert import CXRBertConfig, CXRBertTokenizer
from .modelling_cxrbert import BertProjectionHead, CXRBertModel, CXRBertOutput

__all__ = ["CXRBertConfig", "CXRBertTokenizer", "CXRBertModel", "CXRBertOutput", "BertProjectionHead"]
