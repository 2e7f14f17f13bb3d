"""Per-model registry record (mirror of /root/reference/src/vlrlhf/models/utils.py:18-31).  The SFT / PPO / reward
slots stay in the record so a reference user finds the same twelve fields; they are None on the DPO-only path."""
from dataclasses import dataclass
from typing import Any


@dataclass
class ModelCoreMapper:
    model: Any
    processor: Any
    dpo_collator: Any
    dpo_trainer: Any
    reward_model: Any = None
    value_model: Any = None
    reward_collator: Any = None
    reward_trainer: Any = None
    sft_collator: Any = None
    sft_trainer: Any = None
    ppo_collator: Any = None
    ppo_trainer: Any = None
