"""Stand-in for pytorch_lightning==1.6.5 (env.yml:247) -- TEST INFRASTRUCTURE, build-authored."""
import random

import numpy as np
import torch


class LightningModule(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        self._logged = {}

    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, name, value, **k):
        self._logged[name] = value

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")


class LightningDataModule:
    def __init__(self, *a, **k):
        pass


def seed_everything(seed=0, workers=False):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed
