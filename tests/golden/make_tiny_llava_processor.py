#!/usr/bin/env python3
"""Writes tests/golden/tiny_llava_processor/: a REAL LlamaTokenizerFast (sentencepiece-style BPE with the "▁" normaliser,
BOS, merges that can cross the prompt/answer boundary) trained on a few sentences + a CLIPImageProcessor for 28x28 images,
saved by transformers' own LlavaProcessor.save_pretrained.  Data fixture for the entry-point / tokenize_row tests (no
network, no real checkpoint in the image)."""
import os

from tokenizers import Tokenizer, decoders, models, normalizers, trainers
from transformers import CLIPImageProcessor, LlamaTokenizerFast, LlavaProcessor

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_llava_processor")
CORPUS = ["USER: <image>\nWhat is shown in this picture? ASSISTANT: A small brown dog is running across the green field.",
          "USER: <image>\nDescribe the image in detail. ASSISTANT: The image shows two people sitting at a wooden table with cups of coffee.",
          "the quick brown fox jumps over the lazy dog", "What colour is the car? The car is red and it is parked near the house.",
          "How many apples are on the table? There are three apples and one orange on the table.",
          "Is there a cat in the photo? No, there is no cat, but there is a bird on the fence."] * 4


def main():
    tok = Tokenizer(models.BPE(unk_token="<unk>", fuse_unk=True))
    tok.normalizer = normalizers.Sequence([normalizers.Prepend("▁"), normalizers.Replace(" ", "▁")])
    tok.decoder = decoders.Sequence([decoders.Replace("▁", " "), decoders.Fuse(), decoders.Strip(" ", 1, 0)])
    alphabet = list("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789.,?!:;'\"-<>\n▁")
    tok.train_from_iterator(CORPUS, trainers.BpeTrainer(vocab_size=400, special_tokens=["<unk>", "<s>", "</s>"], initial_alphabet=alphabet))
    t = LlamaTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>", add_bos_token=True,
                           add_eos_token=False, legacy=False)
    t.add_tokens(["<image>", "<pad>"], special_tokens=True)
    ip = CLIPImageProcessor(size={"shortest_edge": 28}, crop_size={"height": 28, "width": 28})
    LlavaProcessor(image_processor=ip, tokenizer=t, patch_size=14, vision_feature_select_strategy="default").save_pretrained(OUT)
    print(sorted(os.listdir(OUT)), len(t), t.convert_tokens_to_ids("<image>"))


if __name__ == "__main__":
    main()
