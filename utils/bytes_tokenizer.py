"""UTF-8 byte tokenizer with 6 reserved ids (PAD BOS EOS MASK CLS SEP), as used by example_language.py."""
import numpy as np


class BytesTokenizer:
    RESERVED = ("pad", "bos", "eos", "mask", "cls", "sep")

    def __init__(self):
        self._num_reserved_tokens = len(self.RESERVED)

    def to_int(self, inputs):
        raw = inputs.encode("utf-8") if isinstance(inputs, str) else inputs
        return (np.frombuffer(raw, dtype=np.uint8).astype(np.int32) + self._num_reserved_tokens).astype(np.int32)

    def to_string(self, inputs: np.ndarray) -> str:
        body = inputs[inputs >= self._num_reserved_tokens] - self._num_reserved_tokens
        return body.astype(np.uint8).tobytes().decode("utf-8", errors="replace")

    @property
    def vocab_size(self) -> int:
        return 256 + self._num_reserved_tokens


for _i, _n in enumerate(BytesTokenizer.RESERVED):
    setattr(BytesTokenizer, f"{_n}_token", property(lambda self, _i=_i: _i))
