"""GlobalInteractor -- MI355X path of models/aggregators/agg_hivt.py:18-58: three layers of
edge-conditioned graph attention over the actors valid at the reference step, then the K-mode fan-out.

Constructor kwargs as in configs/nusargo/hivt_nuSArgo_sdesepenc_sdedec.yml:51-59; call signature
`aggregator(data=data, local_embed=...) -> [K, N, 64]`; state_dict keys as in SURVEY.md App. C.
"""
from trajsde_amd.models.params import ParamTree
from trajsde_amd import runtime


class GlobalInteractor(ParamTree):
    def __init__(self, **kwargs) -> None:
        super().__init__()
        self.set_init_seed(kwargs.pop("init_seed", None))
        for key, value in kwargs.items():
            setattr(self, key, value)
        d = self.embed_dim
        if not self.rotate:
            raise NotImplementedError("rotate=False is not built (the shipped config rotates, CFG:59)")
        if d != 64 or self.num_heads not in (4, 8) or self.edge_dim != 2:
            raise NotImplementedError("kernels are specialised for embed_dim=64, 4 or 8 heads, 2-d edges")
        self.multiple_input_embedding("rel_embed", [self.edge_dim, self.edge_dim], d)
        for i in range(self.num_layers):
            self.attention_block(f"global_interactor_layers.{i}", d,
                                 qkv=("lin_q_node", "lin_k_node", "lin_k_edge", "lin_v_node", "lin_v_edge"))
        self.layernorm("norm", d)
        self.linear("multihead_proj", self.num_modes * d, d)
        self.set_init_seed(None)
        self._rt = runtime.StageRuntime(self, "aggregator")

    def forward(self, data, local_embed, noise=None, prepared=None):
        """`noise` (ours, optional): the NoiseSpec whose dropout key seeds this stage's train-mode dropout masks;
        `prepared` (ours, optional): the handle of `runtime.StageRuntime.prefetch_rel_embed` of this forward"""
        return self._rt.aggregator_forward(data, local_embed, noise, prepared=prepared)
