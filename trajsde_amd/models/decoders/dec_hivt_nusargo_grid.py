"""MLPDecoder -- MI355X path of models/decoders/dec_hivt_nusargo_grid.py:11-63 "GDEC": fuse local + global embeddings,
then two MLP heads emit all T future steps at once (loc, scale) and a three-layer head the mode logits.
Constructor kwargs as in configs/nusargo/hivt_nuSArgo_trmenc_mlpdec.yml:52-60; call signature
`decoder(data=..., local_embed=..., global_embed=...) -> {'loc','pi','reg_mask','local_embed','global_embed'}`.
"""
from trajsde_amd.models.params import ParamTree
from trajsde_amd import runtime


class MLPDecoder(ParamTree):
    def __init__(self, **kwargs) -> None:
        super().__init__()
        self.set_init_seed(kwargs.pop("init_seed", None))
        for key, value in kwargs.items():
            setattr(self, key, value)
        self.input_size, self.hidden_size = self.global_channels, self.local_channels
        d, t = self.hidden_size, self.future_steps
        if d != 64 or self.input_size != 64 or not 0 < t <= 64:
            raise NotImplementedError("kernels are specialised for 64 channels, future_steps <= 64")
        self.linear("aggr_embed.0", d, self.input_size + d)
        self.layernorm("aggr_embed.1", d)
        self.head("loc", d, d, 2 * t)
        if self.uncertain:
            self.head("scale", d, d, 2 * t)
        else:                                                    # dec_hivt_nusargo_grid.py:31: no scale head, 'loc' [K, N, T, 2] (:58-59)
            self.absent_head("scale", d, d, 2 * t)
        self.linear("pi.0", d, d + self.input_size)
        self.layernorm("pi.1", d)
        self.linear("pi.3", d, d)
        self.layernorm("pi.4", d)
        self.linear("pi.6", 1, d)
        self.set_init_seed(None)
        self.num_layers = t                      # the MLP-decoder pack recipe takes the step count in this slot
        self._rt = runtime.StageRuntime(self, "decoder_mlp")

    def forward(self, data, local_embed, global_embed):
        return self._rt.mlp_decoder_forward(data, local_embed, global_embed)
