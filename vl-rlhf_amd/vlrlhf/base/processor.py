"""Abstract processor / chat template (mirror of /root/reference/src/vlrlhf/base/processor.py:11-164)."""
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import List, Literal, Optional, Union

import torch

from ..utils.common import pad_to_length


@dataclass
class VLChatTemplate:
    system_begin: Optional[str]
    system_end: Optional[str]
    user_begin: str
    user_end: str
    assistant_begin: str
    assistant_end: str
    image_placeholder: str


class VLProcessor(ABC):
    @property
    @abstractmethod
    def tokenizer(self):
        raise NotImplementedError

    @property
    @abstractmethod
    def chat_template(self) -> VLChatTemplate:
        raise NotImplementedError

    @property
    @abstractmethod
    def image_processor(self):
        raise NotImplementedError

    @abstractmethod
    def save_pretrained(self, output_dir: str):
        raise NotImplementedError

    @abstractmethod
    def process_batch_conv(self, sources, system_message=None, add_end_for_empty_value=False) -> dict:
        raise NotImplementedError

    @staticmethod
    @abstractmethod
    def format_multimodal_prompt(prompt: str, img_paths: Optional[Union[List[str], str]] = None):
        raise NotImplementedError

    @staticmethod
    @abstractmethod
    def remove_image_placeholder(prompt: str):
        raise NotImplementedError

    @staticmethod
    @abstractmethod
    def is_multimodal_prompt_valid(prompt: str) -> bool:
        raise NotImplementedError

    @staticmethod
    def make_single_turn_conv(prompt: str, answer: str = ""):
        return [{"from": "user", "value": prompt}, {"from": "assistant", "value": answer}]

    @abstractmethod
    def train(self):
        raise NotImplementedError

    @abstractmethod
    def infer(self):
        raise NotImplementedError

    def __call__(self, texts=None, convs=None, images_path=None, padding: bool = True,
                 padding_side: Literal["right", "left"] = "left", check_format: bool = True):
        """tokenize texts or conversations into (optionally padded) tensors (reference :95-164)."""
        if texts is None and convs is None:
            raise ValueError("texts and convs cannot be both None")
        if texts is not None and convs is not None:
            raise ValueError("texts and convs cannot be both set")
        if isinstance(texts, str):
            texts = [texts]
        if texts is not None:
            if images_path is not None:
                texts = [self.format_multimodal_prompt(t, p) for t, p in zip(texts, images_path)]
            if check_format and images_path is not None:
                for t in texts:
                    if not self.is_multimodal_prompt_valid(t):
                        raise ValueError(f"invalid multimodal prompt: {t}")
            enc = [self.tokenizer(t) for t in texts]
            ids = [e["input_ids"] for e in enc]
            masks = [e["attention_mask"] for e in enc]
            labels = None
        else:
            full = self.process_batch_conv(convs)["full"]
            ids, masks, labels = full["input_ids"], full["attention_mask"], full["labels"]
        if not padding:
            return dict(input_ids=ids, attention_mask=masks, labels=labels)
        n = max(len(i) for i in ids)
        pad_id = self.tokenizer.pad_token_id

        def pad(rows, value):
            return torch.stack([pad_to_length(torch.tensor(r, dtype=torch.long), n, value, padding_side=padding_side)
                                for r in rows])

        out = dict(input_ids=pad(ids, pad_id), attention_mask=pad(masks, 0))
        if labels is not None:
            out["labels"] = pad(labels, -100)
        return out
