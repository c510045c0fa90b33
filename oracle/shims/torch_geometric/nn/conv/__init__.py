"""MessagePassing(aggr='add', node_dim=0) restated."""
import inspect

import torch


class MessagePassing(torch.nn.Module):
    _special = {"edge_index", "index", "ptr", "size_i", "size_j", "size", "dim_size"}

    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2, **kwargs):
        super().__init__()
        if aggr != "add" or flow != "source_to_target" or node_dim != 0:
            raise NotImplementedError("stand-in covers aggr='add', source_to_target, node_dim=0")
        self._msg_params = list(inspect.signature(self.message).parameters)
        self._upd_params = list(inspect.signature(self.update).parameters)[1:]

    def propagate(self, edge_index, size=None, **kwargs):
        src, dst = edge_index[0], edge_index[1]
        sizes = [None, None] if size is None else list(size)
        # size inference: a (src_feats, dst_feats) tuple gives both sides (ENC:792); a plain node tensor
        # that is gathered with _i/_j gives a square graph (ENC:609, AGG:131).
        for name in self._msg_params:
            if not name.endswith(("_i", "_j")):
                continue
            v = kwargs.get(name[:-2])
            if isinstance(v, (tuple, list)):
                cand = [v[0].size(0) if torch.is_tensor(v[0]) else None, v[1].size(0)]
            elif torch.is_tensor(v):
                cand = [v.size(0), v.size(0)]
            else:
                continue
            sizes = [sizes[k] if sizes[k] is not None else cand[k] for k in range(2)]
        msg_kwargs = {}
        for name in self._msg_params:
            if name in ("index",):
                msg_kwargs[name] = dst
            elif name == "ptr":
                msg_kwargs[name] = None
            elif name == "size_i":
                msg_kwargs[name] = sizes[1]
            elif name == "size_j":
                msg_kwargs[name] = sizes[0]
            elif name == "edge_index":
                msg_kwargs[name] = edge_index
            elif name.endswith("_i") or name.endswith("_j"):
                base, which = name[:-2], name[-1]
                v = kwargs[base]
                if isinstance(v, (tuple, list)):
                    v = v[1] if which == "i" else v[0]
                msg_kwargs[name] = None if v is None else v.index_select(0, dst if which == "i" else src)
            else:
                msg_kwargs[name] = kwargs.get(name)
        msg = self.message(**msg_kwargs)
        out = msg.new_zeros((sizes[1],) + tuple(msg.shape[1:]))
        out.index_add_(0, dst, msg)
        upd_kwargs = {name: kwargs[name] for name in self._upd_params if name in kwargs}
        return self.update(out, **upd_kwargs)

    def message(self, x_j):
        return x_j

    def update(self, inputs):
        return inputs
