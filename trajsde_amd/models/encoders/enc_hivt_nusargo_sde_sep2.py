"""LocalEncoderSDESepPara2 -- MI355X path of the reference's local encoder
(models/encoders/enc_hivt_nusargo_sde_sep2.py:25-202): per-step agent-agent graph attention,
latent SDE + GRU recurrence over the 21 history steps, agent-lane attention.

Same constructor kwargs (configs/nusargo/hivt_nuSArgo_sdesepenc_sdedec.yml:27-46), same call
signature `encoder(data=data) -> (local_embed, diff_in, diff_out, label_in, label_out)`, same
state_dict keys.  All arithmetic runs in the HIP kernels of
trajsde_amd/csrc through the C-ABI of include/trajsde_hip.h; there is no PyTorch fallback.
"""
from typing import Optional

import torch

from trajsde_amd.models.params import ParamTree
from trajsde_amd import runtime


class LocalEncoderSDESepPara2(ParamTree):
    def __init__(self, **kwargs) -> None:
        super().__init__()
        self.set_init_seed(kwargs.pop("init_seed", None))
        for key, value in kwargs.items():
            setattr(self, key, value)
        if not self.parallel:
            raise NotImplementedError("parallel=False is not implemented (reference: ENC:122-123)")
        if not self.run_backwards or self.method != "euler" or self.adaptive or self.sde_layers != 2:
            raise NotImplementedError("only the shipped solver settings are built: euler, fixed step, "
                                      "run_backwards, sde_layers=2 (CFG:37-46)")
        d, h = self.embed_dim, self.historical_steps
        if d != 64 or self.num_heads != 8 or self.node_dim != 2 or self.edge_dim != 2:
            raise NotImplementedError("kernels are specialised for embed_dim=64, 8 heads, 2-d inputs (CFG:30-33)")
        self.real_label, self.fake_label = 0, 1

        self.token("aa_encoder.bos_token", h, d)
        self.single_input_embedding("aa_encoder.center_embed", self.node_dim, d)
        self.multiple_input_embedding("aa_encoder.nbr_embed", [self.node_dim, self.edge_dim], d)
        self.attention_block("aa_encoder", d)
        self.multiple_input_embedding("al_encoder.lane_embed", [self.node_dim, self.edge_dim], d)
        self.attention_block("al_encoder", d)
        for name, rows in (("is_intersection_embed", 2), ("turn_direction_embed", 3), ("traffic_control_embed", 2)):
            self.token(f"al_encoder.{name}", rows, d)          # present in checkpoints, unused (ENC:724-729)
        for gate in ("update_gate", "reset_gate", "new_state_net"):
            # GRU_Unit draws N(0, 0.1) (ODEU:211-215) when it is built (ENC:49), but the encoder's closing
            # `self.apply(init_weights)` (ENC:64) re-initialises every nn.Linear, these included: xavier-uniform, zero bias
            self.linear(f"gru_unit.{gate}.0", d, 2 * d)
            self.linear(f"gru_unit.{gate}.2", d, d)
        self.sde_nets("lsde_func", d, ("g_nus", "g_argo"))
        self.token("hidden", d)
        self.set_init_seed(None)
        self._rt = runtime.StageRuntime(self, "encoder")

    def forward(self, data, noise: Optional["runtime.NoiseSpec"] = None, preserve_side_effects: Optional[bool] = None):
        """`preserve_side_effects` (default: the `preserve_side_effects` constructor kwarg, else False): also leave
        `data['edge_index_{t}']` / `data['edge_attr_{t}']` on the batch as the reference's forward does (ENC:107-110)."""
        if preserve_side_effects is None:
            preserve_side_effects = bool(getattr(self, "preserve_side_effects", False))
        return self._rt.encoder_forward(data, noise, preserve_side_effects=preserve_side_effects)

    def forward_ood(self, data, noise: Optional["runtime.NoiseSpec"] = None):
        """ENC:204-370: 10 stochastic recurrences from a zero state -> (local_embed, per-actor std)."""
        return self._rt.encoder_forward_ood(data, noise, n_samples=10)
