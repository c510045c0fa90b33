"""trajsde_amd: MI355X-native (gfx950) implementation of TrajSDE's forecasting hot path.

PredictionModelSDENet.forward = local encoder -> global interactor -> SDE decoder
(reference: models/model_base_mix_sde.py:74-102), hand-written HIP behind a C-ABI
(include/trajsde_hip.h, trajsde_amd/csrc), with Python nn.Module stage classes that keep the
reference's YAML plugin boundary, constructor kwargs, call signatures and state_dict keys.
"""
__version__ = "0.1.0"
