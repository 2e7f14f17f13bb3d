"""DDPO token-difference index sets (mirror of /root/reference/src/vlrlhf/utils/diff_lib.py:116-125,158-180).

Host-side integer work on the label ids (difflib), exactly as the reference does it; the product turns the index sets
into a device mask once per batch instead of calling .tolist() inside get_batch_logps."""
import difflib
from typing import List, Tuple

import torch


def get_match_info(a_seq, b_seq, min_match_size=1):
    blocks = difflib.SequenceMatcher(None, a_seq, b_seq).get_matching_blocks()
    kept = [m for m in blocks[:-1] if m[2] >= min_match_size] + [blocks[-1]]
    return [(m[0], m[0] + m[2]) for m in kept], [(m[1], m[1] + m[2]) for m in kept]


def _gaps(matches, length):
    """spans strictly between consecutive kept matches (+ head before the first and tail after the last)."""
    out, prev = [], 0
    for lo, hi in matches:
        out.append((prev, lo))
        prev = hi
    out.append((prev, length))
    return out


def generate_modification_mapping(a_seq, b_seq, min_match_size=3):
    a_m, b_m = get_match_info(a_seq, b_seq, min_match_size)
    mod = {}
    for a_span, b_span in zip(_gaps(a_m, len(a_seq)), _gaps(b_m, len(b_seq))):
        if a_span[0] != a_span[1] and b_span[0] != b_span[1]:      # replace spans only: both sides non-empty
            mod[a_span] = b_span
    return mod


def get_diff_ids(a_seq, b_seq, min_match_size=3) -> Tuple[List[int], List[int]]:
    mod = generate_modification_mapping(a_seq, b_seq, min_match_size)
    a_ids = sorted({i for s in mod.keys() for i in range(*s)})
    b_ids = sorted({i for s in mod.values() for i in range(*s)})
    return a_ids, b_ids


def ddpo_shared_mask(labels: torch.Tensor, label_pad_token_id: int = -100, min_match_size: int = 3) -> torch.Tensor:
    """[2B, S-1] bool mask of the shifted label positions that differ between the chosen and the rejected half
    (reference base/trainer.py:161-184)."""
    sh = labels[:, 1:].detach().to("cpu").clone()
    sh[sh == label_pad_token_id] = 0
    n = sh.shape[0] // 2
    assert n * 2 == sh.shape[0]
    mask = torch.zeros_like(sh, dtype=torch.bool)
    rows = sh.tolist()
    for i in range(n):
        c, r = get_diff_ids(rows[i], rows[n + i], min_match_size)
        mask[i, c] = True
        mask[n + i, r] = True
    return mask
