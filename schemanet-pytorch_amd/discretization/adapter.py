from typing import Tuple

import torch


class Adapter:
    """Strips the cls token before vector quantisation and re-attaches it afterwards
    (reference discretization/visual_word_encoder.py:10-20).  Sequence-first tensors."""

    def __init__(self):
        self.cls_token: torch.Tensor = None

    def adapt(self, x: torch.Tensor) -> torch.Tensor:
        self.cls_token = x[:1]
        return x[1:]

    def reconstruct(self, x: torch.Tensor, match: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return torch.cat((self.cls_token, x), dim=0), match
