"""CPU-only checks: the C-ABI library loads and exports every symbol include/vlr.h declares, argument errors are
reported without a GPU, the host-side mirror (collator, tokenize_row, concatenation, DDPO ids, schedule, flat layout)
reproduces the reference-generated golden vectors, and the product path refuses to run without the MI355X."""
import ctypes
import json
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests.golden_util import GOLDEN, load_case, t

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vlrlhf import _hip
    l = _hip.lib()
    hdr = open(os.path.join(ROOT, "include", "vlr.h")).read()
    declared = set(re.findall(r"\b(vlr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    for name in sorted(declared):
        assert hasattr(l, name), f"{name} declared in include/vlr.h but not exported by libvlr_hip.so"
    for name in _hip.exported_symbols():
        assert name in declared, f"{name} bound in _hip.py but not declared in include/vlr.h"
    assert _hip.helper("vlr_abi_version") == 9


def test_argument_errors_without_gpu():
    from vlrlhf import _hip
    l = _hip.lib()
    assert l.vlr_gemm_bf16(9, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, None) == 1
    assert b"layout" in l.vlr_last_error()
    assert l.vlr_dpo_loss(None, None, None, None, 4, 0.1, 0.0, 7, 0, None, None, None, None, None, None, None, None) == 1
    assert b"Unknown loss type" in l.vlr_last_error()
    assert l.vlr_attn_fwd(None, None, None, 8, None, 8, None, None, 1, 8, 1, 96, 1, 1.0, None) == 1
    assert b"head_dim" in l.vlr_last_error()
    assert _hip.helper("vlr_rmsnorm_bwd_workspace_bytes", 4096) >= 256 * 4096 * 4
    # ABI v7: the two-adapter layer passes and the row-set products validate before they launch
    assert l.vlr_decoder_layer_fwd_lora2(None, None, None, None, None, None, None, 0, 0, None, None, None, None, 1, 8, None) == 1
    assert b"vlr_decoder_layer_fwd_lora2: null argument" in l.vlr_last_error()
    assert l.vlr_gemm_grouped_bits_rows(2, None, None, None, 8, 8, 8, 8, 8, 8, 1, 0, 0, 0, 1.0, 0, 0, 0, 0.0, 0, None, 0, None, None) == 1
    assert b"layout 0 / 1" in l.vlr_last_error()
    assert l.vlr_gemm_grouped_bits_ktiles(0, None, None, None, 8, 8, 8, 8, 8, 8, 1, 0, 0, 0, 1.0, 0, 0, 0, 0.0, 0, None, 0, None, None) == 1
    assert b"layout 2 only" in l.vlr_last_error()
    assert l.vlr_rows_tile_list(None, 8, None, None) == 1
    assert b"vlr_rows_tile_list" in l.vlr_last_error()


def test_product_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vlrlhf import _hip
    from vlrlhf.engine import LlavaHipEngine
    from vlrlhf.base.trainer import VLDPOTrainer
    with pytest.raises(_hip.VlrError):
        LlavaHipEngine(dict(hidden=128, inter=256, vocab=64, layers=1, heads=1, vit_hidden=64, vit_heads=1, vit_mlp=128,
                            vit_layers=2, image_size=28, patch_size=14, image_token=60))
    with pytest.raises(_hip.VlrError):   # CPU logits are rejected, not silently handled by PyTorch
        VLDPOTrainer.get_batch_logps(torch.zeros(2, 4, 8), torch.zeros(2, 4, dtype=torch.long))
    with pytest.raises(ValueError):
        VLDPOTrainer.get_batch_logps(torch.zeros(2, 4, 8), torch.zeros(2, 5, dtype=torch.long))


def test_collator_matches_reference_golden():
    from vlrlhf.base.collator import VLDPODataCollatorWithPadding
    ka = np.load(os.path.join(GOLDEN, "known_answers.npz"))
    rows = json.loads(bytes(ka["collator_rows_json"]).decode())
    got = VLDPODataCollatorWithPadding(pad_token_id=0, label_pad_token_id=-100)(rows)
    n = 0
    for k in ka.files:
        if k.startswith("collator."):
            exp = torch.from_numpy(ka[k])
            g = got[k[len("collator."):]]
            assert torch.equal(g, exp) if exp.dtype == torch.int64 else torch.allclose(g.float(), exp.float()), k
            n += 1
    assert n == 10 and got["img_path"] == ["a.jpg", "b.jpg"]
    for case in ("llava_tiny", "llava_hipsmall"):
        z, cfg, W, W_ref, batch, rws = load_case(case)
        got = VLDPODataCollatorWithPadding()(rws)
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                assert torch.equal(got[k], v), (case, k)


class _Trainer:
    """VLDPOTrainer without a model (host-side methods only)."""
    def __new__(cls, **kw):
        from vlrlhf.base.trainer import VLDPOTrainer
        tr = VLDPOTrainer.__new__(VLDPOTrainer)
        tr.__dict__.update(dict(label_pad_token_id=-100, padding_value=0, max_length=512, max_prompt_length=128,
                                truncation_mode="keep_end", is_encoder_decoder=False, loss_type="sigmoid"), **kw)
        return tr


def test_concatenated_inputs_matches_reference_golden():
    for case in ("llava_tiny", "llava_hipsmall"):
        z, cfg, W, W_ref, batch, rows = load_case(case)
        tr = _Trainer()
        cb = tr.concatenated_inputs(batch)
        for k in ("concatenated_input_ids", "concatenated_attention_mask", "concatenated_labels"):
            assert torch.equal(cb[k], t(z, "cat." + k)), (case, k)
        pv = cb["concatenated_img_input_dict"]["pixel_values"]
        n = batch["img_input_dict"]["pixel_values"].shape[0]
        assert pv.shape[0] == 2 * n and torch.equal(pv[:n], pv[n:]) and pv._vlr_dup == 2
        assert tr.concatenated_inputs(batch)["concatenated_input_ids"] is cb["concatenated_input_ids"]   # memoised
    b = dict(batch)
    b.pop("_vlr_concat")
    b["img_input_dict"] = dict(paths=["a", "b"], bad=3)
    with pytest.raises(ValueError, match="Unsupported type"):
        tr.concatenated_inputs(b)


def test_ddpo_ids_match_reference_golden():
    from vlrlhf.utils.diff_lib import ddpo_shared_mask, get_diff_ids
    ka = np.load(os.path.join(GOLDEN, "known_answers.npz"))
    labels = torch.from_numpy(ka["lp.labels"])
    sh = labels[:, 1:].clone()
    sh[sh == -100] = 0
    c, r = get_diff_ids(sh[0].tolist(), sh[2].tolist(), 3)
    assert c == ka["lp.ddpo_c0"].tolist() and r == ka["lp.ddpo_r0"].tolist()
    for case in ("llava_tiny", "llava_hipsmall"):
        z, cfg, W, W_ref, batch, rows = load_case(case)
        m = ddpo_shared_mask(t(z, "merged_labels"))
        B = batch["chosen_input_ids"].shape[0]
        for b in range(B):
            assert torch.where(m[b])[0].tolist() == z[f"ddpo_chosen_ids_{b}"].tolist()
            assert torch.where(m[B + b])[0].tolist() == z[f"ddpo_rejected_ids_{b}"].tolist()


class FakeTokenizer:
    """whitespace-free character-bigram tokenizer: deterministic, merges across boundaries like BPE can."""
    bos_token_id, eos_token_id, pad_token_id, unk_token = 1, 2, 0, "<unk>"
    bos_token = "<s>"

    def __call__(self, text, add_special_tokens=True, padding=False):
        ids, i = [], 0
        while i < len(text):
            if text.startswith("<image>", i):
                ids.append(90)
                i += 7
            else:
                ids.append(3 + (ord(text[i]) % 80))
                i += 1
        if add_special_tokens:
            ids = [self.bos_token_id] + ids
        return {"input_ids": ids, "attention_mask": [1] * len(ids)}


def test_tokenize_row_template_and_truncation():
    from vlrlhf.models.Llava import LlavaProcessor
    proc = LlavaProcessor(tokenizer=FakeTokenizer())
    conv = proc.make_single_turn_conv(proc.format_multimodal_prompt("What is this?", "x.jpg"), "")
    raw = proc.process_batch_conv([conv], add_end_for_empty_value=False)["raw_str"][0]
    assert raw == "USER: <image>\nWhat is this?ASSISTANT: "          # SURVEY Appendix A.6 (verified on the reference)
    tr = _Trainer(processor=proc, tokenizer=proc.tokenizer, max_length=60, max_prompt_length=20)
    row = tr.tokenize_row(dict(prompt="What is this?", chosen="a cat on a mat", rejected="a dog", img_path="x.jpg"))
    tok = FakeTokenizer()
    p_ids = [1] + tok(raw, add_special_tokens=False)["input_ids"]
    c_ids = tok("a cat on a mat", add_special_tokens=False)["input_ids"] + [2]
    assert row["prompt_input_ids"] == p_ids
    assert row["chosen_input_ids"] == p_ids + c_ids
    assert row["chosen_labels"] == [-100] * len(p_ids) + c_ids
    assert row["rejected_labels"][: len(p_ids)] == [-100] * len(p_ids) and row["rejected_input_ids"][-1] == 2
    assert row["img_path"] == "x.jpg" and row["chosen_input_ids"].count(90) == 1
    # prompt too long -> keep_end to max_prompt_length, then answers to max_length - max_prompt_length
    tr2 = _Trainer(processor=proc, tokenizer=proc.tokenizer, max_length=30, max_prompt_length=12)
    row2 = tr2.tokenize_row(dict(prompt="What is this?", chosen="x" * 40, rejected="y" * 5, img_path="x.jpg"))
    assert row2["prompt_input_ids"] == p_ids[-12:]
    assert len(row2["chosen_input_ids"]) == 12 + 18 and len(row2["rejected_input_ids"]) == 12 + 6
    tr3 = _Trainer(processor=proc, tokenizer=proc.tokenizer, max_length=30, max_prompt_length=12, truncation_mode="nope")
    with pytest.raises(ValueError, match="Unknown truncation mode"):
        tr3.tokenize_row(dict(prompt="What is this?", chosen="x" * 40, rejected="y", img_path="x.jpg"))
    assert LlavaProcessor.format_multimodal_prompt("a <image> b", ["p"]) == "a <image>\n b"
    assert LlavaProcessor.remove_image_placeholder("<image>\nhi") == "hi"


def test_lr_schedule_and_layout():
    from vlrlhf.engine import ParamLayout
    tr = _Trainer(args=SimpleNamespace(learning_rate=1e-5, warmup_ratio=0.1, lr_scheduler_type="cosine"))
    assert tr.lr_at(0, 100) == 0.0 and abs(tr.lr_at(10, 100) - 1e-5) < 1e-12 and abs(tr.lr_at(55, 100) - 0.5e-5) < 1e-9
    assert tr.lr_at(100, 100) < 1e-12
    cfg = dict(hidden=256, inter=512, vocab=320, layers=3, vit_hidden=128)
    lay = ParamLayout(cfg)
    spans = sorted(lay.bucket_after.values())
    assert spans[0][0] == 0 and spans[-1][1] == lay.numel
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert b == c                                   # the DDP buckets tile the flat gradient exactly
    names = [hf for hf, *_ in lay.hf_names()]
    assert len(names) == len(set(names)) == 3 * 9 + 7
    assert lay.n_decay % 8 == 0 and all(o % 8 == 0 for o in lay.offset.values())
    no_decay = [n for n, o in lay.offset.items() if o >= lay.n_decay]
    assert all(("ln" in n or n == "norm" or ".b" in n) for n in no_decay)


def test_registry_surface():
    from vlrlhf.models.Llava import core_mapper
    from vlrlhf.utils.auto_load import MODEL_NICKNAME_MAP, auto_core_mapper
    assert MODEL_NICKNAME_MAP["LlavaForConditionalGeneration"] == "Llava"
    assert auto_core_mapper("LlavaForConditionalGeneration") is core_mapper
    assert core_mapper.dpo_trainer.__mro__[1].__name__ == "VLDPOTrainer"
    with pytest.raises(NotImplementedError):
        auto_core_mapper("InstructBlipForConditionalGeneration")
    import inspect
    from vlrlhf.base.trainer import VLDPOTrainer
    params = list(inspect.signature(VLDPOTrainer.__init__).parameters)[1:]
    assert params[:13] == ["model", "ref_model", "beta", "label_smoothing", "loss_type", "args", "data_collator",
                           "label_pad_token_id", "padding_value", "truncation_mode", "train_dataset", "eval_dataset", "processor"]
    assert len(params) == 32 and params[-1] == "reference_free"


def test_prefetch_loader_order_depth_and_errors():
    """input pipeline: background collation keeps order, runs at most `depth` batches ahead, surfaces
    collator exceptions on the consuming thread and stops its thread when the consumer stops early."""
    import threading
    import time
    from vlrlhf.base.loader import PrefetchLoader
    made = []

    def rows():
        for i in range(6):
            yield [dict(i=i)]

    def collate(r):
        made.append(r[0]["i"])
        return dict(x=torch.full((2,), r[0]["i"]), meta=[r[0]["i"]], img_input_dict=dict(pixel_values=torch.zeros(1, 3, 2, 2)))

    out = []
    for b in PrefetchLoader(rows, collate, None, depth=2):
        time.sleep(0.02)
        out.append(int(b["x"][0]))
        assert len(made) <= len(out) + 3            # queue (2) + the one being collated
    assert out == list(range(6))

    def bad(r):
        if r[0]["i"] == 2:
            raise ValueError("boom")
        return dict(x=torch.zeros(1))
    with pytest.raises(ValueError, match="boom"):
        list(PrefetchLoader(rows, bad, None, depth=2))
    n0 = threading.active_count()
    it = iter(PrefetchLoader(rows, collate, None, depth=1))
    next(it)
    it.close()
    time.sleep(0.3)
    assert threading.active_count() <= n0


def test_ctypes_signatures_match_the_header():
    """every prototype of include/vlr.h has the same number (and pointer / integer / float kind) of parameters as its
    ctypes binding in vlrlhf/_hip.py - a shifted argument would otherwise only show up as garbage on the GPU."""
    import ctypes as C
    from vlrlhf import _hip
    hdr = open(os.path.join(ROOT, "include", "vlr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    protos = dict(re.findall(r"\b(?:int|const char\*)\s+(vlr_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S))

    def kind(param):
        p = " ".join(param.split())
        if p in ("void", ""):
            return None
        if "*" in p or "vlr_stream_t" in p:
            return "p"
        if re.search(r"\b(float|double)\b", p):
            return "f"
        return "i"

    def ckind(ct):
        if ct in (C.c_void_p, C.c_char_p):
            return "p"
        if ct in (C.c_float, C.c_double):
            return "f"
        return "i"

    checked = 0
    for name, sig in list(_hip._SIGS.items()) + list(_hip._INT_HELPERS.items()):
        assert name in protos, name
        want = [k for k in (kind(x) for x in protos[name].split(",")) if k]
        have = [ckind(c) for c in sig]                       # the stream is the last entry of every _SIGS signature
        assert have == want, f"{name}: header {want} vs ctypes {have}"
        checked += 1
    assert checked >= 45


def test_vlfeedback_pair_mining_matches_reference(tmp_path):
    """utils/data.py: the pair mining of `vlfeedback_paired` against the reference's own `make_batch_pairs` (captured by
    oracle/make_golden.py datasets): largest-gap-only (score_margin -1) and margin 1.0, unparsable ratings and all-tie samples."""
    from vlrlhf.utils.data import DATASET_MAP, vlfeedback_pairs
    g = json.load(open(os.path.join(GOLDEN, "vlfeedback_pairs.json")))
    s = g["samples"]
    samples = [dict(prompt=s["prompt"][i], img_path=s["img_path"][i], completions=s["completions"][i]) for i in range(len(s["prompt"]))]
    for margin, exp in g["results"].items():
        rows = vlfeedback_pairs(samples, float(margin) if margin != "-1" else -1)
        got = {k: [r[k] for r in rows] for k in ("prompt", "chosen", "rejected", "img_path")}
        assert got == {k: exp[k] for k in got}, margin
    # the entry through DATASET_MAP on a local export
    p = tmp_path / "vlf.jsonl"
    p.write_text("\n".join(json.dumps(x) for x in samples))
    rows = DATASET_MAP["vlfeedback_paired"](SimpleNamespace(data_path=str(p), image_root="/imgs", score_margin=-1))
    assert len(rows) == len(g["results"]["-1"]["prompt"]) and rows[0]["img_path"].startswith("/imgs/")
    with pytest.raises(RuntimeError, match="no network"):
        DATASET_MAP["vlfeedback_paired"](SimpleNamespace(data_path=None, score_margin=-1))


def test_rccl_channel_bounds_follow_the_cu_reservation(monkeypatch):
    """parallel.rccl_channel_env: NCCL_MAX_NCHANNELS = VLR_COMM_CUS (default 16) in the environment the ranks are launched with, unless
    the user set it (no floor is forced: NCCL_MIN_NCHANNELS stays the user's); VLR_COMM_CUS=0 leaves RCCL alone.  (The library's own
    communicator carries the bound in its configuration - vlr_comm_init_cfg - and does not depend on this environment.)"""
    from vlrlhf import parallel as P
    monkeypatch.delenv("VLR_COMM_CUS", raising=False)
    e = {}
    assert P.rccl_channel_env(e) == ("16", None) and e == {"NCCL_MAX_NCHANNELS": "16"}
    e = {"NCCL_MAX_NCHANNELS": "8", "NCCL_MIN_NCHANNELS": "junk"}
    assert P.rccl_channel_env(e) == ("8", "junk")                   # the user's values win and are never parsed
    monkeypatch.setenv("VLR_COMM_CUS", "24")
    e = {}
    assert P.rccl_channel_env(e) == ("24", None)
    monkeypatch.setenv("VLR_COMM_CUS", "0")
    e = {}
    assert P.rccl_channel_env(e) == (None, None) and e == {}


def test_cus_are_given_up_only_while_buckets_are_in_flight(monkeypatch):
    """GradReducer: vlr_set_comm_cus(k) with the first bucket of a backward, vlr_set_comm_cus(0) at wait() - the forward passes and a step
    without an exchange (gradient-accumulation micro-step, world 1) keep the whole chip; VLR_COMM_CUS_SCOPE=step holds them all along."""
    from vlrlhf import _hip, parallel as P
    calls = []
    monkeypatch.setattr(_hip, "helper", lambda name, *a: calls.append((name,) + a) or 0)
    flat = torch.zeros(64)
    for scope, expect in (("backward", [16, 0, 16, 0]), ("step", [16])):
        monkeypatch.setenv("VLR_COMM_CUS_SCOPE", scope)
        r = P.GradReducer(flat, {"a": (0, 32), "b": (32, 64)})
        assert r.comm_cus == 0 and r.reserve_scope == scope          # world 1 / CPU: nothing reserved
        r.cuda, r.comm_cus = True, 16                                  # the bookkeeping alone, as on a GPU rank of a data-parallel run
        del calls[:]
        if scope == "step":
            r._reserve(True)                                           # (GradReducer.__init__ on a GPU rank)
        for _ in range(2):                                             # two steps
            r._reserve(True); r._reserve(True)                         # buckets "a", "b"
            r._reserve(False)                                          # wait()
        assert calls == [("vlr_set_comm_cus", k) for k in expect]


def test_sampling_filter_follows_the_logits_warpers():
    """LlavaForRL.generate's temperature / top-k / top-p filter against a plain restatement of transformers' warpers"""
    from vlrlhf.models.Llava import sampling_filter
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(5, 40, generator=g) * 3
    out = sampling_filter(logits, 0.7, 7, 1.0)
    assert bool(((out > float("-inf")).sum(-1) == 7).all())
    keep = (logits / 0.7).topk(7, dim=-1).indices
    assert bool(torch.isfinite(out.gather(1, keep)).all())
    out = sampling_filter(logits, 1.0, 0, 0.8)
    for b in range(5):
        p = logits[b].softmax(-1)
        order = p.argsort(descending=True)
        csum = p[order].cumsum(0)
        n_keep = int((csum < 0.8).sum()) + 1                       # the smallest prefix whose mass reaches top_p
        want = torch.zeros(40, dtype=torch.bool)
        want[order[:n_keep]] = True
        assert torch.equal(torch.isfinite(out[b]), want), b
    assert bool(torch.isfinite(sampling_filter(logits, 1.0, 1, 0.01)).sum(-1).eq(1).all())       # one token always survives


def test_sampling_filter_matches_the_hf_warpers():
    """... and against transformers' OWN TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper run in the build container
    (oracle/make_golden_sampling.py -> tests/golden/sampling_warpers.json): the same tokens survive (ties included) with the same scores"""
    import json
    from vlrlhf.models.Llava import sampling_filter
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sampling_warpers.json")))
    logits = torch.tensor(G["logits"], dtype=torch.float32)
    assert len(G["cases"]) >= 8
    for c in G["cases"]:
        out = sampling_filter(logits.clone(), c["temperature"], c["top_k"], c["top_p"])
        kept = torch.tensor(c["kept"], dtype=torch.bool)
        assert torch.equal(torch.isfinite(out), kept), (c["temperature"], c["top_k"], c["top_p"])
        want = torch.tensor([[v if v is not None else 0.0 for v in row] for row in c["scores"]], dtype=torch.float32)
        assert torch.allclose(out[kept], want[kept], rtol=1e-6, atol=1e-6), (c["temperature"], c["top_k"], c["top_p"])


def _stub_trainer(n_batches, ga, **args):
    """VLDPOTrainer.train's control flow without a model: a stub engine, `n_batches` micro-batches per epoch, checkpoints recorded"""
    from types import SimpleNamespace
    from vlrlhf.base.trainer import VLDPOTrainer

    class Eng:
        master, reducer = object(), None

        def zero_grad(self): pass
        def optimizer_step(self, **kw): pass
        def grad_norm(self): return 0.0

    class T(VLDPOTrainer):
        def __init__(self):      # noqa: super().__init__ needs a real model
            self.args = SimpleNamespace(gradient_accumulation_steps=ga, num_train_epochs=1.0, max_steps=-1, logging_steps=1000, save_strategy="epoch",
                                        save_steps=500, evaluation_strategy="no", **args)
            self.model = SimpleNamespace(engine=Eng())
            self.precompute_ref_log_probs = False
            self.eval_dataset = None
            self.state = SimpleNamespace(global_step=0)
            self.saved = []

        def _batches_per_epoch(self): return n_batches
        def get_train_batches(self, epoch, skip=0): return iter(range(skip, n_batches))
        def training_step(self, model, batch): return torch.zeros(())
        def prefetch_reference(self, b): return b
        def log(self, logs): return logs
        def save_checkpoint(self, step, micro, epoch, window_len=0): self.saved.append((step, micro, epoch))

    return T()


def test_epoch_save_strategy_always_leaves_a_final_checkpoint():
    """save_strategy="epoch" (HF fires on_epoch_end - and saves - at the end of the epoch training stops in, whether it was consumed or cut short):
    a run whose batch count is not a multiple of gradient_accumulation_steps, one that is, and one that max_steps ends mid-epoch all write
    a checkpoint at their last optimizer step, exactly once"""
    t = _stub_trainer(10, 4)                   # 10 micro-batches, windows of 4: 2 optimizer steps, 2 micro-batches left over
    t.train()
    assert t.saved == [(2, 8, 0)], t.saved
    t = _stub_trainer(8, 4)
    t.train()
    assert t.saved == [(2, 8, 0)], t.saved
    t = _stub_trainer(8, 2)
    t.args.max_steps = 3                       # stops inside the epoch
    t.train()
    assert t.saved == [(3, 6, 0)], t.saved
    t = _stub_trainer(6, 4)                    # two epochs: the first ends inside a window, its checkpoint is written at the next optimizer step
    t.args.num_train_epochs = 2.0
    t.train()
    assert [s for s, _, _ in t.saved] == [2] or [s for s, _, _ in t.saved] == [1, 2], t.saved
    assert t.saved[-1][0] == 2
