"""`TextInput`: tokenisation front-end (host side, stays Python) — same behaviour as the reference's
`health_multimodal/text/data/io.py:17-58`: wrap a single string, reject special tokens (except [MASK]), strip
trailing '!?.' and `batch_encode_plus(add_special_tokens=True, padding='longest', return_tensors='pt')`."""
import logging
from typing import Any, List, Union

TypePrompts = Union[str, List[str]]

logger = logging.getLogger(__name__)


class TextInput:
    """Text input class for inference and deployment.

    :param tokenizer: A BertTokenizer-like object (`batch_encode_plus`, `all_special_tokens`, `mask_token`,
        `convert_ids_to_tokens`).
    """

    def __init__(self, tokenizer: Any) -> None:
        self.tokenizer = tokenizer

    def tokenize_input_prompts(self, prompts: TypePrompts, verbose: bool) -> Any:
        prompts = [prompts] if isinstance(prompts, str) else prompts
        self.assert_special_tokens_not_present(" ".join(prompts))

        prompts = [prompt.rstrip("!?.") for prompt in prompts]  # removes punctuation from end of prompt
        tokenizer_output = self.tokenizer.batch_encode_plus(batch_text_or_text_pairs=prompts,
                                                            add_special_tokens=True,
                                                            padding='longest',
                                                            return_tensors='pt')
        if verbose:
            for prompt in tokenizer_output.input_ids:
                input_tokens = self.tokenizer.convert_ids_to_tokens(prompt.tolist())
                logger.info(f"Input tokens: {input_tokens}")

        return tokenizer_output

    def assert_special_tokens_not_present(self, prompt: str) -> None:
        """Check if the input prompts contain special tokens."""
        special_tokens = list(self.tokenizer.all_special_tokens)
        if self.tokenizer.mask_token in special_tokens:
            special_tokens.remove(self.tokenizer.mask_token)  # [MASK] is allowed
        if any(map(lambda token: token in prompt, special_tokens)):
            raise ValueError(f"The input \"{prompt}\" contains at least one special token ({special_tokens})")
