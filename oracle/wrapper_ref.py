"""Oracle (TEST INFRASTRUCTURE): functional restatement of the neighbor-encoding wrapper.

Reference: /root/reference/model/modelling_cross_attention.py
  TextPooler               :879-893
  get_text_embs            :978-1004
  get_visual_embs          :1006-1027
  forward (interleave)     :1038-1114
and /root/reference/model/modelling_self_attention.py :282-332 (self-attention-concat fusion),
/root/reference/model/graph.py :17-31 (GCN).

The frozen encoders themselves (RoBERTa / CLIP) are third-party ``transformers`` modules in the
reference too; the oracle takes their outputs (``last_hidden_state`` / ``pooler_output``) as inputs.
"""
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .lm_ref import LMConfig, causal_lm_forward


def text_pooler(p: Dict[str, torch.Tensor], last_hidden_state: torch.Tensor) -> torch.Tensor:
    """CLS token -> Linear -> tanh (:888-893)."""
    return torch.tanh(F.linear(last_hidden_state[:, 0], p["text_pooler.dense.weight"], p["text_pooler.dense.bias"]))


def project_neighbors(p: Dict[str, torch.Tensor], name: str, pooled: torch.Tensor, pos_ids: Optional[torch.Tensor],
                      batch: int, n_tokens: int) -> torch.Tensor:
    """pooled [B*N, h_enc] -> Linear(h_enc -> n_tokens*d) (+ position embedding) -> [B, N, n_tokens, d]
    (:997-1004, :1020-1027).  ``name`` is "text" or "visual"."""
    e = F.linear(pooled, p[f"{name}_embeddings.weight"], p[f"{name}_embeddings.bias"])
    if pos_ids is not None and f"{name}_position_embeddings.weight" in p:
        e = e + F.embedding(pos_ids.reshape(-1), p[f"{name}_position_embeddings.weight"])
    return e.reshape(batch, -1, n_tokens, e.shape[-1] // n_tokens)


def interleave_neighbors(text_embeds: torch.Tensor, visual_embeds: torch.Tensor, text_pos_ids: torch.Tensor,
                         visual_pos_ids: torch.Tensor, text_locations: torch.Tensor,
                         image_locations: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Scatter text / image neighbor tokens into slot order and build the key-valid mask (:1080-1104).
    text_embeds [B,Nt,n,d], visual_embeds [B,Ni,n,d] -> ([B,(Nt+Ni)*n,d], bool [B,(Nt+Ni)*n])."""
    B, Nt, n, d = text_embeds.shape
    Ni = visual_embeds.shape[1]
    out = torch.zeros(B, Nt + Ni, n, d, dtype=text_embeds.dtype)
    valid = torch.zeros(B, Nt + Ni, n, dtype=torch.bool)
    for b in range(B):
        for j in range(Nt):
            out[b, text_locations[b, j]] = text_embeds[b, j]
            valid[b, text_locations[b, j]] = bool(text_pos_ids[b, j] > 0)
        for j in range(Ni):
            out[b, image_locations[b, j]] = visual_embeds[b, j]
            valid[b, image_locations[b, j]] = bool(visual_pos_ids[b, j] > 0)
    return out.reshape(B, (Nt + Ni) * n, d), valid.reshape(B, (Nt + Ni) * n)


def cross_attention_model_forward(p: Dict[str, torch.Tensor], cfg: LMConfig, batch: Dict[str, torch.Tensor],
                                  text_last_hidden: Optional[torch.Tensor], visual_pooled: Optional[torch.Tensor],
                                  context: str, n_tokens: int):
    """CrossAttentionModel.forward (:1038-1114) given the frozen encoders' outputs.
    ``p`` holds the wrapper's state dict (``lm.`` prefix for the LM)."""
    lm = {k[3:]: v for k, v in p.items() if k.startswith("lm.")}
    B = batch["input_ids"].shape[0]
    ne, nm = None, None
    if context == "text_only":
        pooled = text_pooler(p, text_last_hidden)
        te = project_neighbors(p, "text", pooled, batch["neighbor_pos_ids"], B, n_tokens)
        ne = te.reshape(B, -1, te.shape[-1])
        nm = (batch["neighbor_pos_ids"] > 0).repeat_interleave(n_tokens, dim=1)
    elif context in ("section_all", "all"):
        pooled = text_pooler(p, text_last_hidden)
        te = project_neighbors(p, "text", pooled, batch["neighbor_pos_ids"], B, n_tokens)
        ve = project_neighbors(p, "visual", visual_pooled, batch["neighbor_images_pos_ids"], B, n_tokens)
        ne, nm = interleave_neighbors(te, ve, batch["neighbor_pos_ids"], batch["neighbor_images_pos_ids"],
                                      batch["text_locations"], batch["image_locations"])
    return causal_lm_forward(lm, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], ne, nm)


def gcn_forward(w1: torch.Tensor, w2: torch.Tensor, X: torch.Tensor, adj: torch.Tensor) -> torch.Tensor:
    """2-layer GCN over a dense adjacency with a prepended zero root (graph.py:17-31); no biases."""
    X = torch.cat([torch.zeros_like(X[:, :1]), X], dim=1)
    X = F.relu(F.linear(torch.cat([X, adj @ X], dim=-1), w1))
    X = F.linear(torch.cat([X, adj @ X], dim=-1), w2)
    return X[:, 1:]


def self_attention_concat_inputs(embed_tokens: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                                 labels: torch.Tensor, neighbor_embeds: torch.Tensor, neighbor_valid: torch.Tensor,
                                 lpe_proj: Optional[torch.Tensor] = None):
    """Self-attention fusion (modelling_self_attention.py:305-330): neighbor tokens are appended AFTER the
    token embeddings, the mask is extended and labels padded with -100.  ``lpe_proj`` = already-projected
    Laplacian PE [B, 1+N, n, d] whose root row is dropped (:311-315)."""
    B = input_ids.shape[0]
    if lpe_proj is not None:
        neighbor_embeds = neighbor_embeds + lpe_proj[:, 1:].reshape(B, -1, neighbor_embeds.shape[-1])
    x = torch.cat([F.embedding(input_ids, embed_tokens), neighbor_embeds], dim=1)
    m = torch.cat([attention_mask.to(torch.float32), neighbor_valid.to(torch.float32)], dim=1)
    lab = torch.cat([labels, torch.full(neighbor_valid.shape, -100, dtype=labels.dtype)], dim=1)
    return x, m, lab
