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
        """raw texts or formatted conversations -> padded input_ids / attention_mask / labels (reference base/processor.py:95-164): a text
        becomes the single-turn conversation [user: text, assistant: ""] and goes through process_batch_conv like a conversation does; when
        images are passed, a text without the image placeholder gets it through format_multimodal_prompt (with the reference's warning).
        Pinned by tests/golden/processor_answers.json["call"] (the reference's own method on the same tokenizer files).  Beyond the
        reference: a single string is accepted for `texts`, and padding=False returns the un-padded lists (the reference leaves its
        result unbound there)."""
        if texts is not None and convs is not None:
            raise AssertionError("You can only pass texts or convs, not both.")
        texts = [texts] if isinstance(texts, str) else (list(texts) if texts else None)
        if texts and images_path is not None and check_format:
            # prompts that lack the image placeholder get it; one warning for the batch
            lacking = [not self.is_multimodal_prompt_valid(t) for t in texts]
            if any(lacking):
                import warnings
                texts = [self.format_multimodal_prompt(t, img) if bad else t for t, img, bad in zip(texts, images_path, lacking)]
                warnings.warn("You passed images, but your prompts are not in multimodal format. The image placeholder is added to "
                              "them automatically; prepare multimodal prompts in advance.")
        batch_conv = [self.make_single_turn_conv(text) for text in texts] if texts else convs
        if batch_conv is None:
            raise ValueError("texts and convs cannot be both None")
        full = self.process_batch_conv(batch_conv)["full"]
        ids, masks, labels = full["input_ids"], full["attention_mask"], full["labels"]
        if not padding:
            return dict(input_ids=ids, attention_mask=masks, labels=labels)
        n = max(len(i) for i in ids)

        def pad(rows, value):
            return torch.stack([pad_to_length(torch.tensor(r, dtype=torch.long), n, value, padding_side=padding_side) for r in rows])

        return dict(input_ids=pad(ids, self.tokenizer.pad_token_id), attention_mask=pad(masks, 0), labels=pad(labels, -100))
