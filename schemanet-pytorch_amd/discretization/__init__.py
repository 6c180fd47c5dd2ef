"""Drop-in for the reference's `discretization` package (nearest-visual-word assignment).

Kept API (reference discretization/__init__.py:5-6, discretization.py:10-81,
visual_word_encoder.py:10-20): `Discretization`, `Adapter`.  Added: `DiscretizationModule`, an
nn.Module with the call contract and attribute path of the TorchScript artefact
`discretization-jit.pth` that IngredientModelWrapper consumes.
`VisualWordEncoder` (a forward-hook helper used only by the codebook evaluation) is out of the
hot path and is not provided.  `discretization.kmeans` is the GPU form of the codebook extraction
(reference scripts/extract_ingredients.py: scipy k-means on the collected patch tokens).
"""
from .discretization import Discretization, DiscretizationModule
from .adapter import Adapter

__all__ = ["Discretization", "DiscretizationModule", "Adapter"]
