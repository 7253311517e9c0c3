"""Host-side tokenisation front-end of the text path (stays Python; the kernels start at token ids).

Behavioural contract of the reference's `TextInput` (`health_multimodal/text/data/io.py:17-58`), restated rather than
transcribed:
  * a single prompt is treated as a batch of one;
  * a prompt that spells out a tokenizer special token ([CLS], [SEP], [PAD], [UNK] ...) is refused with ValueError —
    the mask token is the one exception, because `predict_masked_tokens` needs it;
  * sentence-final '!', '?' and '.' are dropped before encoding;
  * encoding = `batch_encode_plus` with special tokens added, padded to the longest prompt, PyTorch tensors.
"""
import logging
from typing import Any, Iterable, List, Sequence, Union

TypePrompts = Union[str, List[str]]
_TRAILING_PUNCTUATION = "!?."

logger = logging.getLogger(__name__)


def _as_batch(prompts: TypePrompts) -> List[str]:
    return [prompts] if isinstance(prompts, str) else list(prompts)


class TextInput:
    """Turns raw prompt strings into the `input_ids` / `attention_mask` pair the text encoder consumes."""

    def __init__(self, tokenizer: Any) -> None:
        self.tokenizer = tokenizer

    def _reserved_tokens(self) -> Sequence[str]:
        allowed = getattr(self.tokenizer, "mask_token", None)
        return [tok for tok in self.tokenizer.all_special_tokens if tok != allowed]

    def assert_special_tokens_not_present(self, prompt: str) -> None:
        reserved = self._reserved_tokens()
        found = [tok for tok in reserved if tok in prompt]
        if found:
            raise ValueError(f"The input \"{prompt}\" contains at least one special token ({reserved}): {found}")

    def _log_tokens(self, rows: Iterable[Any]) -> None:
        for row in rows:
            logger.info("Input tokens: %s", self.tokenizer.convert_ids_to_tokens(row.tolist()))

    def tokenize_input_prompts(self, prompts: TypePrompts, verbose: bool) -> Any:
        batch = _as_batch(prompts)
        self.assert_special_tokens_not_present(" ".join(batch))
        cleaned = [text.rstrip(_TRAILING_PUNCTUATION) for text in batch]
        encoded = self.tokenizer.batch_encode_plus(batch_text_or_text_pairs=cleaned, add_special_tokens=True,
                                                   padding="longest", return_tensors="pt")
        if verbose:
            self._log_tokens(encoded.input_ids)
        return encoded
