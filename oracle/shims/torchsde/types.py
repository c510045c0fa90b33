"""Stand-in for torchsde.types (type aliases only; sdeint.py:12)."""
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union  # noqa: F401

import torch

Tensor = torch.Tensor
Tensors = Sequence[Tensor]
TensorOrTensors = Union[Tensor, Tensors]
Scalar = Union[float, Tensor]
Vector = Union[Sequence[float], Tensor]
Module = torch.nn.Module
Modules = Sequence[Module]
ModuleOrModules = Union[Module, Modules]
Size = torch.Size
Sizes = Sequence[Size]
