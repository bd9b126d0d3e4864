"""Graph positional-encoding producers the reference calls but never defines (wikiweb2m/data.py:434 `utils.compute_LPE`,
:438 `utils.normalize_graph`; its README defers to another repository).  Only the CONSUMER shapes are pinned by the
reference: lpe [1+Nt+Ni, k] with k = 1+Nt+Ni-5 (model/modelling_self_attention.py:137-139), graph [1+Nt+Ni, 1+Nt+Ni]
(model/graph.py:17-31).  Definitions chosen here (parity UNPINNED):
  compute_LPE     : eigenvectors of the symmetric-normalised Laplacian L = I - D^-1/2 A D^-1/2 of the undirected page
                    graph, ascending eigenvalue order, the trivial first eigenvector dropped, k columns, sign fixed
                    so that each eigenvector's largest-magnitude entry is positive (deterministic), isolated nodes -> 0.
  normalize_graph : D^-1/2 (A + I) D^-1/2 (the GCN propagation matrix).
"""
import torch


def dense_adjacency(edge_index: torch.Tensor, node_num: int) -> torch.Tensor:
    A = torch.zeros(node_num, node_num, dtype=torch.float64)
    if edge_index.numel():
        src, dst = edge_index[0].long(), edge_index[1].long()
        ok = (src < node_num) & (dst < node_num) & (src != dst)
        A[src[ok], dst[ok]] = 1.0
        A[dst[ok], src[ok]] = 1.0
    return A


def normalize_graph(A: torch.Tensor) -> torch.Tensor:
    A = A.to(torch.float64) + torch.eye(A.shape[0], dtype=torch.float64)
    dinv = A.sum(1).clamp_min(1e-12).rsqrt()
    return (dinv[:, None] * A * dinv[None, :]).to(torch.float32)


def compute_LPE(edge_index: torch.Tensor, node_num: int, k: int = None) -> torch.Tensor:
    k = node_num - 5 if k is None else k
    A = dense_adjacency(edge_index, node_num)
    deg = A.sum(1)
    dinv = torch.where(deg > 0, deg.clamp_min(1e-12).rsqrt(), torch.zeros_like(deg))
    L = torch.eye(node_num, dtype=torch.float64) - dinv[:, None] * A * dinv[None, :]
    evals, evecs = torch.linalg.eigh(L)
    vecs = evecs[:, 1:1 + k]
    if vecs.shape[1] < k:
        vecs = torch.cat([vecs, torch.zeros(node_num, k - vecs.shape[1], dtype=torch.float64)], dim=1)
    idx = vecs.abs().argmax(dim=0)
    sign = torch.sign(vecs[idx, torch.arange(vecs.shape[1])])
    sign[sign == 0] = 1.0
    vecs = vecs * sign[None, :]
    vecs[deg == 0] = 0.0
    return vecs.to(torch.float32)
