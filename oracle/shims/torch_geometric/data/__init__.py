"""Data / Batch / Dataset / DataLoader stand-ins (attribute + item access, index offsetting on collate)."""
import torch


class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None, **kwargs):
        object.__setattr__(self, "_store", {})
        for k, v in dict(x=x, edge_index=edge_index, edge_attr=edge_attr, y=y, pos=pos).items():
            if v is not None:
                self._store[k] = v
        for k, v in kwargs.items():
            if v is not None or k == "num_nodes":
                self._store[k] = v
        if self._store.get("num_nodes", 0) is None:
            del self._store["num_nodes"]

    # attribute and item access are interchangeable (MODEL:76-85, ENC:107-110)
    def __getattr__(self, key):
        store = object.__getattribute__(self, "_store")
        if key in store:
            return store[key]
        if key == "num_nodes":
            x = store.get("x")
            return None if x is None else x.size(0)
        if key in ("x", "y", "edge_index", "edge_attr", "pos"):
            return None
        raise AttributeError(key)

    def __setattr__(self, key, value):
        self._store[key] = value

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        self._store[key] = value

    def __delitem__(self, key):
        del self._store[key]

    def __delattr__(self, key):
        del self._store[key]

    def __contains__(self, key):
        return key in self._store

    @property
    def keys(self):
        return [k for k, v in self._store.items() if v is not None]

    def __inc__(self, key, value, *args, **kwargs):
        return self.num_nodes if "index" in key or key == "face" else 0

    def __cat_dim__(self, key, value, *args, **kwargs):
        return -1 if "index" in key or key == "face" else 0

    def to(self, device):
        for k, v in self._store.items():
            if torch.is_tensor(v):
                self._store[k] = v.to(device)
        return self


class Batch(Data):
    @classmethod
    def from_data_list(cls, data_list, follow_batch=None, exclude_keys=None):
        out = cls.__new__(cls)
        object.__setattr__(out, "_store", {})
        keys = [k for k in data_list[0].keys if k != "num_nodes"]
        cum = {k: 0 for k in keys}
        chunks = {k: [] for k in keys}
        batch_vec, n_tot = [], 0
        for i, d in enumerate(data_list):
            n = d.num_nodes
            for k in keys:
                v = d[k]
                if torch.is_tensor(v):
                    inc = d.__inc__(k, v)
                    if torch.is_tensor(inc) or inc != 0:
                        v = v + cum[k]
                    cum[k] = cum[k] + inc
                chunks[k].append(v)
            batch_vec.append(torch.full((n,), i, dtype=torch.long))
            n_tot += n
        for k in keys:
            v0 = chunks[k][0]
            if torch.is_tensor(v0) and v0.dim() > 0:
                out._store[k] = torch.cat(chunks[k], dim=data_list[0].__cat_dim__(k, v0))
            elif torch.is_tensor(v0):
                out._store[k] = torch.stack(chunks[k])
            elif isinstance(v0, (int, float)):
                out._store[k] = torch.tensor(chunks[k])
            else:
                out._store[k] = chunks[k]
        out._store["batch"] = torch.cat(batch_vec)
        out._store["num_nodes"] = n_tot
        out._store["num_graphs"] = len(data_list)
        return out


class Dataset(torch.utils.data.Dataset):
    def __init__(self, root=None, transform=None, pre_transform=None, pre_filter=None):
        self.root, self.transform = root, transform
        if hasattr(self, "_download"):
            self._download()
        if hasattr(self, "_process"):
            self._process()

    def __len__(self):
        return self.len()

    def __getitem__(self, idx):
        return self.get(idx)


class DataLoader(torch.utils.data.DataLoader):
    def __init__(self, dataset, batch_size=1, shuffle=False, **kw):
        kw.pop("collate_fn", None)
        super().__init__(dataset, batch_size, shuffle, collate_fn=Batch.from_data_list, **kw)
