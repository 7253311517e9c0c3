"""Prompt dictionaries — same function names and template strings as the reference's `DataRetrieval.py:183-237`
(the strings are data: they fix what CXR-BERT embeds).  The CheXpert CSV/JPEG dataset classes of that file are
outside the hot path (no CheXpert offline); synthetic tensors are used instead."""
from typing import Dict, List

CHEXPERT_COMPETITION_CLASSES = ["Atelectasis", "Cardiomegaly", "Consolidation", "Edema", "Pleural Effusion"]  # Trainer.py:205


def basic_create_prompts(class_list: List[str]) -> Dict[str, Dict[str, List[str]]]:
    """One positive and one negative prompt per class (`DataRetrieval.py:183-198`)."""
    print("*** Basic Prompting ***")
    return {c: {"positive": [f"Findings suggesting {c}"], "negative": [f"No evidence of {c}"]} for c in class_list}


def create_prompts(class_list: List[str], new_prompts: bool = False, train_logit_diff=None):
    """Four positive and four negative prompt templates per class (`DataRetrieval.py:201-237`)."""
    if new_prompts:
        raise NotImplementedError("the MedCLIP-style prompt generator (new_texts_prompts.py) is outside the hot path")
    print("*** Multiple Prompting ***")
    pos = ("Findings consistent with {}", "Findings suggesting {}", "This opacity can represent {}",
           "Findings are most compatible with {}")
    neg = ("There is no {}", "No evidence of {}", "No evidence of acute {}", "No signs of {}")
    return {c: {"positive": [t.format(c) for t in pos], "negative": [t.format(c) for t in neg]} for c in class_list}
