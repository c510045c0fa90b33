"""SDEDecoder -- MI355X path of models/decoders/dec_hivt_nusargo_sde.py:14-105: fuse local+global
embeddings into y0 [K*N,64], Euler-Maruyama over the future grid with learned drift and
scalar-broadcast diffusion, heads -> loc/scale/pi.

Constructor kwargs as in configs/nusargo/hivt_nuSArgo_sdesepenc_sdedec.yml:64-76; call signature
`decoder(data=..., local_embed=..., global_embed=...) -> {'loc','pi','reg_mask'}`; `uncertain: False` (DEC:56, DEC:100-101):
no `scale.*` parameters and `loc` [K, N, T, 2] -- the kernels run with a zero stand-in head and the scale channels are dropped.
"""
from typing import Optional

from trajsde_amd.models.params import ParamTree
from trajsde_amd import runtime


class SDEDecoder(ParamTree):
    def __init__(self, **kwargs) -> None:
        super().__init__()
        self.set_init_seed(kwargs.pop("init_seed", None))
        for key, value in kwargs.items():
            setattr(self, key, value)
        self.input_size, self.hidden_size = self.global_channels, self.local_channels
        d = self.hidden_size
        if d != 64 or self.input_size != 64 or self.method != "euler":
            raise NotImplementedError("kernels are specialised for 64 channels, euler (CFG:64-76)")
        self.linear("aggr_embed.0", d, self.input_size + d)
        self.layernorm("aggr_embed.1", d)
        self.sde_nets("lsde_func", d, ("g_func",))
        self.head("decoder", d, d, 2)
        if self.uncertain:
            self.head("scale", d, d, 2)
        else:                                                    # DEC:56: no scale head; 'loc' is [K, N, T, 2] (DEC:100-101)
            self.absent_head("scale", d, d, 2)
        self.head("pi", d + self.input_size, d, 1)
        self.token("hidden", d)                                  # present in checkpoints, unused (DEC:69)
        self.set_init_seed(None)
        self._rt = runtime.StageRuntime(self, "decoder")

    def forward(self, data, local_embed, global_embed, noise: Optional["runtime.NoiseSpec"] = None):
        return self._rt.decoder_forward(data, local_embed, global_embed, noise)
