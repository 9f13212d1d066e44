"""Typing aliases under the names the reference exports (torchsde/types.py:18-32), for code that does
``from torchsde.types import ...``."""
from typing import Any, Callable, Dict, Optional, Sequence, Tuple, Union  # noqa: F401

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
