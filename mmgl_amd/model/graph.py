"""2-layer GCN positional encoder over a dense normalised adjacency (mirrors reference model/graph.py:6-31).
The two bias-free projections run on the MFMA linear kernel; the [B,N,N] aggregation is a tiny torch bmm."""
import torch
import torch.nn as nn

from .. import ops


class GCN(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dim):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.hidden_dim = hidden_dim
        self.w1 = nn.Linear(2 * input_dim, hidden_dim, bias=False)
        self.w2 = nn.Linear(2 * hidden_dim, output_dim, bias=False)

    def forward(self, X, adj):
        """X [B,N,input_dim] neighbor embeddings, adj [B,N+1,N+1] (node 0 = the null root) -> [B,N,output_dim]."""
        X = torch.cat([X.new_zeros(X.shape[0], 1, X.shape[2]), X], dim=1)
        adj = adj.to(X.dtype)
        X = ops.linear(torch.cat([X, torch.bmm(adj, X)], dim=-1), self.w1.weight, None, act="relu")
        X = ops.linear(torch.cat([X, torch.bmm(adj, X)], dim=-1), self.w2.weight, None)
        return X[:, 1:, :]
