"""FlashSR engine front (placeholder until the device model lands in this round)."""
import torch


def ensure_ready():
    raise RuntimeError("FlashSR device model is not built yet in this revision")


def infer_spans(x_ct: torch.Tensor, n_chunks: int, win: int, hop: int, lowpass: bool) -> torch.Tensor:
    ensure_ready()
