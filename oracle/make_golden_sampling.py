#!/usr/bin/env python3
"""Golden vectors for LlavaForRL.generate's sampling filter.  TEST INFRASTRUCTURE - NOT THE PRODUCT.

The reference trainer samples through transformers' GenerationMixin (/root/reference/src/vlrlhf/base/trainer.py:310-360 ->
`model.generate(..., do_sample=True)`), whose filter is the chain TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper.
This script runs transformers' OWN warper classes (the installed package, in the build container) on seeded logits and stores inputs,
parameters and the filtered scores; tests/test_cabi_and_host.py::test_sampling_filter_matches_the_hf_warpers replays them against
vlrlhf.models.Llava.sampling_filter.

    python oracle/make_golden_sampling.py        # -> tests/golden/sampling_warpers.json
"""
import json
import os

import torch
import transformers
from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(1.0, 50, 1.0), (0.7, 7, 1.0), (1.0, 0, 0.8), (1.3, 5, 0.5), (0.5, 0, 0.95), (1.0, 1, 0.01), (2.0, 64, 0.3), (1.0, 3, 0.999)]


def main():
    g = torch.Generator().manual_seed(1234)
    logits = (torch.randn(6, 64, generator=g) * 3).round(decimals=3)        # 3 decimals: the JSON text is the exact fp32 value
    logits[1, :8] = logits[1, 0]                                            # ties at the top
    logits[2] = torch.linspace(-4, 4, 64)
    ids = torch.zeros(6, 1, dtype=torch.long)
    out = []
    for temperature, top_k, top_p in CASES:
        s = logits.clone()
        if temperature != 1.0:
            s = TemperatureLogitsWarper(temperature)(ids, s)
        if top_k and top_k > 0:
            s = TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=1)(ids, s)
        if top_p < 1.0:
            s = TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=1)(ids, s)
        out.append(dict(temperature=temperature, top_k=top_k, top_p=top_p, kept=torch.isfinite(s).int().tolist(),
                        scores=[[(v if v > -1e30 else None) for v in row] for row in s.tolist()]))
    path = os.path.join(ROOT, "tests", "golden", "sampling_warpers.json")
    with open(path, "w") as f:
        json.dump(dict(transformers=transformers.__version__, chain="TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper (min_tokens_to_keep=1)",
                       logits=logits.tolist(), cases=out), f)
    print("wrote", path)


if __name__ == "__main__":
    main()
