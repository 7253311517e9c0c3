"""Configuration and tokenizer names of CXR-BERT (the public surface of the reference's
`health_multimodal/text/model/configuration_cxrbert.py:11-27`, which Hub checkpoints refer to by class name)."""
from transformers import BertConfig, BertTokenizer


class CXRBertConfig(BertConfig):
    """A BERT configuration with one extra field: `projection_size`, the width of the joint image-text space (128 in BioViL)."""

    model_type = "cxr-bert_encoder"

    def __init__(self, projection_size=128, **bert_kwargs):
        BertConfig.__init__(self, **bert_kwargs)
        self.projection_size = int(projection_size)


class CXRBertTokenizer(BertTokenizer):
    """BertTokenizer under the class name the checkpoint's `tokenizer_config.json` asks for; nothing is overridden."""
