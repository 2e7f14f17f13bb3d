"""Mirror of /root/reference/src/vlrlhf/utils/common.py:58-87 (pad_to_length); the ZeRO-3 save helpers there are out
of scope (no DeepSpeed)."""
from typing import Literal, Union

import torch


def pad_to_length(tensor: torch.Tensor, length: int, pad_value: Union[int, float], dim: int = -1,
                  padding_side: Literal["right", "left"] = "right") -> torch.Tensor:
    if tensor.size(dim) >= length:
        return tensor
    shape = list(tensor.shape)
    shape[dim] = length - tensor.size(dim)
    pad = torch.full(shape, pad_value, dtype=tensor.dtype, device=tensor.device)
    if padding_side == "right":
        return torch.cat([tensor, pad], dim=dim)
    if padding_side == "left":
        return torch.cat([pad, tensor], dim=dim)
    raise ValueError(f"Unknown padding_side: {padding_side}")


def flatten_list(x):
    out = []
    for i in x:
        if isinstance(i, (list, tuple)):
            out.extend(flatten_list(i))
        else:
            out.append(i)
    return out
