"""vlrlhf - MI355X-native DPO training step behind the VL-RLHF surface (vlrlhf.dpo, VLDPOTrainer, per-model wrappers).

Mirror of the reference package layout for the DPO hot path only; the arithmetic runs in
libvlr_hip.so (include/vlr.h) - there is no PyTorch or CPU fallback."""
__version__ = "0.1.0"
