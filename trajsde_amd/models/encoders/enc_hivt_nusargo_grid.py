"""LocalEncoder -- MI355X path of the vanilla HiVT encoder (models/encoders/enc_hivt_nusargo_grid.py:18-93 "GENC"):
per-step agent-agent attention, a causal transformer over each actor's 21 history tokens (TemporalEncoder
GENC:225-292), agent-lane attention.  Constructor kwargs as in configs/nusargo/hivt_nuSArgo_trmenc_mlpdec.yml:24-37,
call signature `encoder(data=data) -> local_embed [N,64]`, state_dict keys as the reference's (the transformer layers
carry nn.TransformerEncoder / nn.MultiheadAttention key names, the causal mask is the `attn_mask` buffer).
All arithmetic runs in the HIP kernels (csrc/attn.hip, csrc/grid.hip); there is no PyTorch fallback.
"""
import torch

from trajsde_amd.models.params import ParamTree
from trajsde_amd import runtime


class LocalEncoder(ParamTree):
    def __init__(self, **kwargs) -> None:
        super().__init__()
        self.set_init_seed(kwargs.pop("init_seed", None))
        for key, value in kwargs.items():
            setattr(self, key, value)
        d, h = self.embed_dim, self.historical_steps
        if not self.parallel:
            raise NotImplementedError("parallel=False is not implemented (reference: GENC:69-75)")
        if d != 64 or self.num_heads not in (4, 8) or self.node_dim != 2 or self.edge_dim != 2 or h != 21:
            raise NotImplementedError("kernels are specialised for embed_dim=64, 4 or 8 heads, 2-d inputs, 21 history steps")
        if not self.input_diff:
            raise NotImplementedError("input_diff=False (no bos tokens) is not built; the shipped config sets it")
        self.token("aa_encoder.bos_token", h, d)
        self.single_input_embedding("aa_encoder.center_embed", self.node_dim, d)
        self.multiple_input_embedding("aa_encoder.nbr_embed", [self.node_dim, self.edge_dim], d)
        self.attention_block("aa_encoder", d)
        t = "temporal_encoder"
        for i in range(self.num_temporal_layers):
            l = f"{t}.transformer_encoder.layers.{i}"
            # init_weights' nn.MultiheadAttention branch (UTIL:114-120): fan_in = fan_out = embed_dim, not the [3d, d] shape
            self.add_param(f"{l}.self_attn.in_proj_weight", self._uniform((3 * d, d), (6.0 / (2 * d)) ** 0.5))
            self.add_param(f"{l}.self_attn.in_proj_bias", torch.zeros(3 * d))
            self.linear(f"{l}.self_attn.out_proj", d, d)
            self.linear(f"{l}.linear1", 4 * d, d)
            self.linear(f"{l}.linear2", d, 4 * d)
            self.layernorm(f"{l}.norm1", d)
            self.layernorm(f"{l}.norm2", d)
        self.layernorm(f"{t}.transformer_encoder.norm", d)
        self.token(f"{t}.padding_token", h, 1, d)
        self.token(f"{t}.cls_token", 1, 1, d)
        self.token(f"{t}.pos_embed", h + 1, 1, d)
        mask = torch.zeros(h + 1, h + 1).masked_fill(~torch.tril(torch.ones(h + 1, h + 1, dtype=torch.bool)), float("-inf"))
        self._leaf_parent(f"{t}.attn_mask")[0].register_buffer("attn_mask", mask)          # GENC:233-234, 251-255
        self.multiple_input_embedding("al_encoder.lane_embed", [self.node_dim, self.edge_dim], d)
        self.attention_block("al_encoder", d)
        for name, rows in (("is_intersection_embed", 2), ("turn_direction_embed", 3), ("traffic_control_embed", 2)):
            self.token(f"al_encoder.{name}", rows, d)          # present in checkpoints, unused (GENC:328-333)
        self.set_init_seed(None)
        self._rt = runtime.StageRuntime(self, "encoder_grid")

    def forward(self, data, noise=None):
        """`noise` (ours): the NoiseSpec whose key the train-mode dropout masks are cut from (csrc/dropout.hpp); eval mode needs none"""
        return self._rt.encoder_grid_forward(data, noise)
