"""utils.softmax / utils.subgraph restated."""
import torch


def subgraph(subset, edge_index, edge_attr=None, relabel_nodes=False, num_nodes=None, return_edge_mask=False):
    """Keep the columns of edge_index whose both endpoints are in `subset` (bool mask); no relabelling."""
    if subset.dtype != torch.bool:
        n = num_nodes if num_nodes is not None else int(edge_index.max()) + 1
        m = torch.zeros(n, dtype=torch.bool, device=edge_index.device)
        m[subset] = True
        subset = m
    if relabel_nodes:
        raise NotImplementedError
    edge_mask = subset[edge_index[0]] & subset[edge_index[1]]
    edge_index = edge_index[:, edge_mask]
    edge_attr = edge_attr[edge_mask] if edge_attr is not None else None
    if return_edge_mask:
        return edge_index, edge_attr, edge_mask
    return edge_index, edge_attr


def softmax(src, index=None, ptr=None, num_nodes=None, dim=0):
    """Per-target segment softmax: exp(src - max_seg) / (sum_seg + 1e-16)."""
    if ptr is not None or dim != 0:
        raise NotImplementedError
    n = int(num_nodes) if num_nodes is not None else (int(index.max()) + 1 if index.numel() else 0)
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    seg_max = src.new_full((n,) + tuple(src.shape[1:]), float("-inf"))
    seg_max = seg_max.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    out = (src - seg_max.gather(0, idx)).exp()
    seg_sum = src.new_zeros((n,) + tuple(src.shape[1:])).scatter_add_(0, idx, out)
    return out / (seg_sum.gather(0, idx) + 1e-16)
