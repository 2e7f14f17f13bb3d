from .collator import VLDPODataCollatorWithPadding  # noqa: F401
from .processor import VLChatTemplate, VLProcessor  # noqa: F401
from .trainer import VLDPOTrainer  # noqa: F401
