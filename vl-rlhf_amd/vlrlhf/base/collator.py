"""DPO collator (mirror of /root/reference/src/vlrlhf/base/collator.py:8-68)."""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch


@dataclass
class VLDPODataCollatorWithPadding:
    r"""Pads the tokenized rows of a DPO batch: chosen_/rejected_ on the right, prompt_ on the left;
    ids -> pad_token_id, labels -> label_pad_token_id, masks -> 0; `*_logps` become a float tensor (cached reference
    log-probs); every other key is passed through as a list."""

    pad_token_id: int = 0
    label_pad_token_id: int = -100
    is_encoder_decoder: Optional[bool] = False
    processor: Optional[Any] = None

    def __call__(self, features: List[Dict[str, Any]]) -> Dict[str, Any]:
        if self.is_encoder_decoder:
            raise NotImplementedError("encoder-decoder models are not on the MI355X DPO path")
        out: Dict[str, Any] = {}
        for k in features[0].keys():
            if k.endswith("_input_ids") or k.endswith("_attention_mask") or k.endswith("_labels"):
                if k.endswith("_input_ids"):
                    pad = self.pad_token_id
                elif k.endswith("_labels"):
                    pad = self.label_pad_token_id
                elif k.endswith("_attention_mask"):
                    pad = 0
                else:
                    raise ValueError(f"Unexpected key in batch '{k}'")
                n = max(len(f[k]) for f in features)
                t = torch.full((len(features), n), pad, dtype=torch.long)
                left = "prompt" in k
                for i, f in enumerate(features):
                    v = torch.as_tensor(f[k], dtype=torch.long)
                    if left:
                        t[i, n - v.numel():] = v
                    else:
                        t[i, : v.numel()] = v
                out[k] = t
            elif k.endswith("_logps"):
                out[k] = torch.tensor([f[k] for f in features])
            else:
                out[k] = [f[k] for f in features]
        return out
