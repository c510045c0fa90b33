"""Stand-in for torch_geometric==2.2.0 (env.yml:286) -- TEST INFRASTRUCTURE, build-authored.

Restates only the semantics TrajSDE's hot path relies on:
Data/Batch containers, MessagePassing(aggr='add', node_dim=0), utils.softmax, utils.subgraph.
"""
from . import data, nn, typing, utils  # noqa: F401
