from typing import Optional, Tuple, Union

import torch

Adj = Union[torch.Tensor, "SparseTensor"]  # noqa: F821
OptTensor = Optional[torch.Tensor]
Size = Optional[Tuple[int, int]]
PairTensor = Tuple[torch.Tensor, torch.Tensor]
