"""A1 (tokenize_row) with a REAL LlamaTokenizerFast - tests/golden/tiny_llava_processor, a sentencepiece-style BPE with BOS
and merges that can cross the prompt/answer boundary.  The processor half is pinned against the reference's own
LlavaProcessor run on the same files (tests/golden/processor_answers.json, written by oracle/make_golden.py processor);
the trl half (tokenize_row / build_tokenized_answer, not vendored in the reference) is pinned through its defining
properties on that tokenizer."""
import json
import os

import pytest

from tests.golden_util import GOLDEN, TINY_PROCESSOR


@pytest.fixture(scope="module")
def proc():
    from vlrlhf.models.Llava import LlavaProcessor
    p = LlavaProcessor(TINY_PROCESSOR)
    p.train()
    return p


def _trainer(proc, **kw):
    from vlrlhf.base.trainer import VLDPOTrainer
    tr = VLDPOTrainer.__new__(VLDPOTrainer)
    tr.__dict__.update(dict(label_pad_token_id=-100, padding_value=0, max_length=512, max_prompt_length=128, truncation_mode="keep_end",
                            is_encoder_decoder=False, loss_type="sigmoid", processor=proc, tokenizer=proc.tokenizer), **kw)
    return tr


def test_processor_matches_the_reference_processor(proc):
    ans = json.load(open(os.path.join(GOLDEN, "processor_answers.json")))
    assert proc.tokenizer.pad_token_id == ans["pad_token_id"] == proc.tokenizer.unk_token_id     # processor.train(): pad = unk
    for a in ans["rows"]:
        r = a["row"]
        prompt = proc.format_multimodal_prompt(r["prompt"], r["img_path"])
        assert prompt == a["formatted_prompt"]
        conv = proc.make_single_turn_conv(prompt, "")
        assert conv == a["conv"]
        pr = proc.process_batch_conv([conv], system_message=None, add_end_for_empty_value=False)
        assert pr["raw_str"][0] == a["prompt_raw_str"] and pr["full"] == a["prompt_full"]
        full = proc.process_batch_conv([proc.make_single_turn_conv(prompt, r["chosen"])])
        assert full["full"] == a["chosen_full"] and full["raw_str"][0] == a["chosen_raw_str"]
        assert proc.is_multimodal_prompt_valid(prompt) == a["valid"] and proc.remove_image_placeholder(prompt) == a["stripped"]


def test_processor_call_matches_the_reference_call(proc):
    """VLProcessor.__call__(texts= / convs=, images_path, padding_side) against the reference's own base method on the same tokenizer files
    (reference base/processor.py:95-164): texts ride make_single_turn_conv -> process_batch_conv, un-formatted prompts get the placeholder"""
    import warnings
    from vlrlhf.base.processor import VLProcessor
    ans = json.load(open(os.path.join(GOLDEN, "processor_answers.json")))
    assert len(ans["call"]) == 5
    for c in ans["call"]:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            enc = VLProcessor.__call__(proc, **c["kwargs"])
        for k in ("input_ids", "attention_mask", "labels"):
            assert enc[k].tolist() == c[k], (c["kwargs"].keys(), k)
        if "texts" in c["kwargs"] and "images_path" in c["kwargs"]:
            assert any("multimodal format" in str(x.message) for x in w)
    with pytest.raises(AssertionError):
        VLProcessor.__call__(proc, texts=["a"], convs=[[]])


def test_tokenize_row_on_a_real_tokenizer(proc):
    ans = json.load(open(os.path.join(GOLDEN, "processor_answers.json")))
    tok = proc.tokenizer
    tr = _trainer(proc)
    img = tok.convert_tokens_to_ids("<image>")
    for a in ans["rows"]:
        r = a["row"]
        row = tr.tokenize_row(dict(r))
        raw = a["prompt_raw_str"]                       # what the reference hands to trl as `prompt` (base/trainer.py:118)
        for side in ("chosen", "rejected"):
            joint = tok(raw + r[side], add_special_tokens=False)["input_ids"]
            ids, lab, am = row[f"{side}_input_ids"], row[f"{side}_labels"], row[f"{side}_attention_mask"]
            # trl's invariant: BOS + enc(prompt + answer) + EOS, whatever the tokenizer merges at the boundary
            assert ids == [tok.bos_token_id] + joint + [tok.eos_token_id]
            n = len(row["prompt_input_ids"]) if side == "chosen" else lab.count(-100)
            assert lab[:n] == [-100] * n and lab[n:] == ids[n:] and -100 not in lab[n:]
            assert am == [1] * len(ids) and ids.count(img) == 1
        assert row["prompt_input_ids"][0] == tok.bos_token_id
        assert row["chosen_input_ids"][: len(row["prompt_input_ids"])] == row["prompt_input_ids"]
        assert row["img_path"] == r["img_path"]


def test_build_tokenized_answer_boundary_merge(proc):
    """the branch of trl's build_tokenized_answer that moves the split one token left when enc(prompt) is not a prefix of
    enc(prompt + answer): 'ASSISTANT: ' ends in a space the BPE merges with the first answer word."""
    tok = proc.tokenizer
    tr = _trainer(proc)
    hit = 0
    for prompt, answer in (("the quick brown", " fox jumps"), ("What colour is the c", "ar is red"), ("USER: hi ASSISTANT: ", "a dog"),
                           ("there is a b", "ird on the fence"), ("on the ta", "ble")):
        full = tok(prompt + answer, add_special_tokens=False)["input_ids"]
        p_alone = tok(prompt, add_special_tokens=False)["input_ids"]
        if len(p_alone) > len(full):       # the merge made the joint encoding SHORTER than the prompt alone: trl raises
            with pytest.raises(ValueError, match="should have the same length"):
                tr.build_tokenized_answer(prompt, answer)
            continue
        out = tr.build_tokenized_answer(prompt, answer)
        assert out["prompt_input_ids"] + out["input_ids"] == full
        if p_alone != full[: len(p_alone)]:
            hit += 1
            assert len(out["prompt_input_ids"]) == len(p_alone) - 1
        else:
            assert out["prompt_input_ids"] == p_alone
    assert hit >= 1, "no example exercised the boundary-merge branch; extend the list"


def test_truncation_with_real_tokens(proc):
    tr = _trainer(proc, max_length=40, max_prompt_length=16)
    row = tr.tokenize_row(dict(prompt="How many apples are on the table? " * 3, chosen="There are three apples. " * 6, rejected="one", img_path="a.jpg"))
    assert len(row["prompt_input_ids"]) == 16 and len(row["chosen_input_ids"]) == 16 + 24
    assert row["chosen_labels"][:16] == [-100] * 16
