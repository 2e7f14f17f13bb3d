"""Deterministic stand-in for the Qwen tokenizer (the real one needs tiktoken + the qwen.tiktoken vocabulary, neither is in the
container): the interface QwenVLProcessor / QwenVLDPOTrainer.tokenize_row use, with the image-slot encoding of
/root/reference/src/vlrlhf/models/QwenVL/tokenization_qwen.py:283-294.  Shared by oracle/make_golden_qwenvl.py (which drives the
REFERENCE's processor / tokenize_row with it to write tests/golden/qwenvl_tokenize.json) and by the tests that replay the fixture."""
import re
import types


class StandInTokenizer:
    """text is split on the special strings; ordinary text -> one id per character (ord % 200 + 256); "<img>path</img>" -> <img>,
    the utf-8 bytes of the path, <imgpad> up to 256 slots, </img>; specials -> their ids."""
    im_start_id, im_end_id, eod_id = 151644, 151645, 151643
    img_start_id, img_end_id, img_pad_id = 151857, 151858, 151859
    SPECIAL = {"<|im_start|>": 151644, "<|im_end|>": 151645, "<|endoftext|>": 151643}
    _SPLIT = re.compile(r"(<\|im_start\|>|<\|im_end\|>|<\|endoftext\|>|<img>.*?</img>)")

    def __init__(self):
        self.pad_token_id = self.eod_id
        self.eos_token_id = self.eod_id
        self.padding_side = "right"

    def __call__(self, text):
        ids = []
        for part in self._SPLIT.split(text):
            if not part:
                continue
            if part in self.SPECIAL:
                ids.append(self.SPECIAL[part])
            elif part.startswith("<img>") and part.endswith("</img>"):
                b = list(part[5:-6].encode("utf-8"))
                ids += [self.img_start_id] + b + [self.img_pad_id] * (256 - len(b)) + [self.img_end_id]
            else:
                ids += [ord(c) % 200 + 256 for c in part]
        return types.SimpleNamespace(input_ids=ids, attention_mask=[1] * len(ids))


class StandInInternLMTokenizer:
    """stand-in for the InternLM2 sentencepiece tokenizer (its model file is not in the container): one id per character, the chat
    markers and <ImageHere> as single ids, BOS prepended when add_special_tokens; returns dicts like a HF tokenizer"""
    SPECIAL = {"[UNUSED_TOKEN_146]": 92543, "[UNUSED_TOKEN_145]": 92542, "<ImageHere>": 92544, "<s>": 1, "</s>": 2}
    _SPLIT = re.compile(r"(\[UNUSED_TOKEN_146\]|\[UNUSED_TOKEN_145\]|<ImageHere>|<s>|</s>)")
    pad_token_id, eos_token_id, bos_token_id = 2, 2, 1

    def __init__(self):
        self.padding_side = "right"

    def __call__(self, text, padding=False, add_special_tokens=True):
        ids = [1] if add_special_tokens else []
        for part in self._SPLIT.split(text):
            if part:
                ids += [self.SPECIAL[part]] if part in self.SPECIAL else [ord(c) % 5000 + 300 for c in part]
        return dict(input_ids=ids, attention_mask=[1] * len(ids))
