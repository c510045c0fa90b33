"""Import and run the reference's hot-path modules from /root/reference over the stand-ins in
oracle/shims (TEST INFRASTRUCTURE; works only where /root/reference exists, i.e. the build container --
never on the GPU box).  Nothing of the reference is copied: its files are executed where they lie.
"""
import contextlib
import copy
import os
import sys

import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = os.environ.get("TRAJSDE_REFERENCE_ROOT", "/root/reference")
REF_CFG = "configs/nusargo/hivt_nuSArgo_sdesepenc_sdedec.yml"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "model_base_mix_sde.py"))


def _install_paths():
    for p in (REFERENCE_ROOT, os.path.join(HERE, "shims"), HERE):
        if p in sys.path:
            sys.path.remove(p)
    # shims first (torchsde, torch_geometric, ...), then oracle/ (noise_source), then the reference tree
    sys.path[:0] = [os.path.join(HERE, "shims"), HERE, REFERENCE_ROOT]


@contextlib.contextmanager
def reference_cwd():
    """Stage files are loaded by relative path (model_base_mix_sde.py:39-41): run with cwd = reference root."""
    old = os.getcwd()
    os.chdir(REFERENCE_ROOT)
    try:
        yield
    finally:
        os.chdir(old)


def load_reference_cfg(num_modes=None, future_steps=None, max_fut_t=None, dataset=None):
    with open(os.path.join(REFERENCE_ROOT, REF_CFG)) as f:
        cfg = yaml.safe_load(f)
    cfg = copy.deepcopy(cfg)
    if num_modes is not None:
        cfg["model_specific"]["kwargs"]["num_modes"] = num_modes
        cfg["aggregator"]["kwargs"]["num_modes"] = num_modes
        cfg["decoder"]["kwargs"]["num_modes"] = num_modes
    if future_steps is not None:
        cfg["model_specific"]["kwargs"]["future_steps"] = future_steps
        cfg["decoder"]["kwargs"]["future_steps"] = future_steps
    if max_fut_t is not None:
        cfg["decoder"]["kwargs"]["max_fut_t"] = max_fut_t
    if dataset is not None:
        cfg["model_specific"]["kwargs"]["dataset"] = dataset
        for m in cfg["metric_args"]:
            m["dataset"] = dataset
    return cfg


def build_reference_model(cfg, seed=0):
    """PredictionModelSDENet(**cfg) exactly as train.py:49-50 / test.py:48-49 build it."""
    _install_paths()
    from importlib.machinery import SourceFileLoader
    with reference_cwd():
        torch.manual_seed(seed)
        ms = cfg["model_specific"]
        mod = SourceFileLoader(ms["module_name"], ms["file_path"]).load_module(ms["module_name"])
        model = getattr(mod, ms["module_name"])(**cfg)
    model.eval()
    return model


def reference_model_class(cfg_path: str = REF_CFG):
    """the reference's model CLASS (not an instance) as train.py:49 / test.py:48 resolve it -- for its static helpers"""
    _install_paths()
    from importlib.machinery import SourceFileLoader
    with open(os.path.join(REFERENCE_ROOT, cfg_path)) as f:
        ms = yaml.safe_load(f)["model_specific"]
    with reference_cwd():
        mod = SourceFileLoader(ms["module_name"], ms["file_path"]).load_module(ms["module_name"])
    return getattr(mod, ms["module_name"])


def to_reference_data(batch):
    """Wrap a trajsde_amd.data.TemporalData (or dict) into the reference's own TemporalData class."""
    _install_paths()
    with reference_cwd():
        from models.utils.util import TemporalData as RefTemporalData
    d = batch.as_dict() if hasattr(batch, "as_dict") else dict(batch)
    out = RefTemporalData()
    for k, v in d.items():
        out[k] = v.clone() if torch.is_tensor(v) else v
    return out


@contextlib.contextmanager
def injected_randn_like():
    """Route torch.randn_like (enc_hivt_nusargo_sde_sep2.py:95) through the oracle noise source."""
    from noise_source import SOURCE
    orig = torch.randn_like

    def patched(t, **kw):
        return SOURCE.standard_normal(t.shape, dtype=t.dtype, device=t.device, tag="randn_like")

    torch.randn_like = patched
    try:
        yield
    finally:
        torch.randn_like = orig


def run_reference_forward(model, batch, seed=None, replay=None):
    """forward(data) with every random draw served (and recorded) by oracle/noise_source.py."""
    _install_paths()
    from noise_source import SOURCE
    SOURCE.reset(seed=seed, replay=replay)
    data = to_reference_data(batch)
    with reference_cwd(), injected_randn_like(), torch.no_grad():
        out = model(data)
    return out, data, list(SOURCE.record)


@contextlib.contextmanager
def injected_dropout(masks):
    """Serve every train-mode dropout of the reference (nn.Dropout.forward -> F.dropout) from `masks`, a list of
    {0, 1/(1-p)} factor tensors in the order the reference's forward calls dropout (per attention block: attention weights
    [E, heads], out_proj output [R, 64], FFN hidden [R, 256], FFN output [R, 64]; blocks: AAEncoder, ALEncoder, the global
    layers).  Eval-mode calls pass through untouched.  Yields the list of shapes served (for the caller's bookkeeping)."""
    import torch.nn.functional as F
    queue, served = list(masks), []
    orig = F.dropout

    def patched(input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        if not queue:
            raise RuntimeError(f"dropout mask queue exhausted at a call on {tuple(input.shape)}")
        m = queue.pop(0)
        if tuple(m.shape) != tuple(input.shape):
            raise RuntimeError(f"dropout mask {tuple(m.shape)} does not fit the reference's tensor {tuple(input.shape)} (call {len(served)})")
        served.append(tuple(input.shape))
        return input * m.to(input.dtype)

    F.dropout = patched
    try:
        yield served
    finally:
        F.dropout = orig
    if queue:
        raise RuntimeError(f"{len(queue)} dropout masks were never asked for: the reference draws dropout in a different order")
