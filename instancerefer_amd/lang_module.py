"""LangModule — drop-in for the reference's models/lang_module.py:7-108 (same ctor signature, state-dict
keys `gru.* word_projection.* fc_a fc_cls fc_rel fc_scene lang_cls.0`, and data_dict keys).

GloVe(300) -> MLP 300->256->256 -> 2-layer (bi)GRU(128) -> four attention heads that pool the
*MLP-projected* embeddings (softmax over padded positions, then mask + renormalise; reference
lang_module.py:61-83) -> 4 x 256-d sentence vectors + 18-way classifier. Dense work: PyTorch-ROCm
(own persistent GRU recurrence kernel on HIP devices, GEMMs on MFMA); the four heads are evaluated as one batched GEMM.
"""
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

import os as _os

from .dense import LangPoolFn, gru_packed

_FUSED_POOL = _os.environ.get('IRX_FUSED_LANG_POOL', '1') != '0'   # dev / test switch: 0 = the heads through ATen (cat, matmul, softmax, bmm)


class LangModule(nn.Module):
    def __init__(self, num_text_classes, use_lang_classifier=True, use_bidir=False, emb_size=300,
                 hidden_size=256):
        super().__init__()
        self.num_text_classes = num_text_classes
        self.use_lang_classifier = use_lang_classifier
        self.use_bidir = use_bidir
        self.gru = nn.GRU(input_size=256, hidden_size=hidden_size, num_layers=2, batch_first=True,
                          bidirectional=self.use_bidir)
        h_dim = 256
        self.word_projection = nn.Sequential(nn.Linear(emb_size, h_dim), nn.ReLU(), nn.Dropout(0.1),
                                             nn.Linear(h_dim, h_dim), nn.ReLU())
        o_dim = 128 * (1 + self.use_bidir)
        self.fc_a = nn.Linear(o_dim, 1)
        self.fc_cls = nn.Linear(o_dim, 1)
        self.fc_rel = nn.Linear(o_dim, 1)
        self.fc_scene = nn.Linear(o_dim, 1)
        if use_lang_classifier:
            self.lang_cls = nn.Sequential(nn.Linear(256, num_text_classes))

    def rnn_encoding(self, embed, length, data_dict):
        if embed.is_cuda:
            # HIP path: persistent GRU recurrence kernel (csrc/irx_gru.hip) + GEMM projections; one D2H of max(len)
            t_max = int(length.max().item()) if "lang_len_max" not in data_dict else int(data_dict["lang_len_max"])
            # Only the first max(len) token positions reach the GRU and the attention pooling (the reference projects
            # all 126 padded rows, lang_module.py:100-102, and drops the rest when it packs the sequence): projecting
            # just those is the same function with the same gradients and a quarter of the MLP's GEMM work at 30 tokens.
            # (dense.mlp_relu2 runs this Sequential as one fused C++ node; at 16 x 30 tokens = 480 rows its 64-row FMA tiles are
            #  slower on the GPU than rocBLAS and the main stream's chain paces the forward: 5.98 vs 5.64 ms per step, three
            #  alternating pairs — so the module itself stays)
            embed = self.word_projection(embed[:, :t_max])
            feats = gru_packed(self.gru, embed, length, t_max)     # (B, T_max, o_dim), zeros at t >= len
        else:
            # host tensors (CPU unit tests of the head logic): the reference's own formulation
            embed = self.word_projection(embed)
            len_cpu = length.detach().to("cpu", torch.int64)
            feats = pack_padded_sequence(embed, len_cpu, batch_first=True, enforce_sorted=False)
            feats, _ = self.gru(feats)
            feats, _ = pad_packed_sequence(feats, batch_first=True)  # (B, T_max, o_dim)
        data_dict['lang_feat'] = feats
        t_max = feats.shape[1]
        if feats.is_cuda and _FUSED_POOL and embed.shape[1] >= t_max:
            # the four heads in one launch each way (csrc/irx_match.hip): order attr, cls, rel, scene. As a C++ node whose pooled
            # vectors are four separate tensors (heads.lang_pool) when the nodes module is there, else dense.LangPoolFn
            from . import heads
            got = heads.lang_pool(self, feats, embed[:, :t_max], length)
            if got is not None:
                att, vecs = got
                data_dict['atten_attr'] = att[:, :, 0]
                data_dict['atten_rel'] = att[:, :, 2]
                data_dict['atten_scene'] = att[:, :, 3]
                data_dict['lang_attr_feats'], data_dict['lang_cls_feats'], data_dict['lang_rel_feats'], data_dict['lang_scene_feats'] = vecs
                return data_dict
            att, pooled = LangPoolFn.apply(feats, embed[:, :t_max], length, self.fc_a.weight, self.fc_a.bias, self.fc_cls.weight,
                                           self.fc_cls.bias, self.fc_rel.weight, self.fc_rel.bias, self.fc_scene.weight,
                                           self.fc_scene.bias)
            data_dict['atten_attr'] = att[:, :, 0]
            data_dict['atten_rel'] = att[:, :, 2]
            data_dict['atten_scene'] = att[:, :, 3]
            data_dict['lang_attr_feats'] = pooled[:, 0]
            data_dict['lang_cls_feats'] = pooled[:, 1]
            data_dict['lang_rel_feats'] = pooled[:, 2]
            data_dict['lang_scene_feats'] = pooled[:, 3]
            return data_dict
        mask = (torch.arange(t_max, device=feats.device).unsqueeze(0) <
                length.to(feats.device).unsqueeze(1)).to(feats.dtype)
        # the four heads as one (o_dim x 4) projection; order: attr, cls, rel, scene
        w = torch.cat([self.fc_a.weight, self.fc_cls.weight, self.fc_rel.weight, self.fc_scene.weight], 0)
        b = torch.cat([self.fc_a.bias, self.fc_cls.bias, self.fc_rel.bias, self.fc_scene.bias], 0)
        att = torch.softmax(feats.matmul(w.t()) + b, dim=1)       # softmax over ALL T_max positions
        att = att * mask.unsqueeze(2)
        att = att / att.sum(1, keepdim=True)                      # (B, T_max, 4)
        pooled = torch.bmm(att.transpose(1, 2), embed[:, :t_max])  # (B, 4, 256)
        data_dict['atten_attr'] = att[:, :, 0]
        data_dict['atten_rel'] = att[:, :, 2]
        data_dict['atten_scene'] = att[:, :, 3]
        data_dict['lang_attr_feats'] = pooled[:, 0]
        data_dict['lang_cls_feats'] = pooled[:, 1]
        data_dict['lang_rel_feats'] = pooled[:, 2]
        data_dict['lang_scene_feats'] = pooled[:, 3]
        return data_dict

    def forward(self, data_dict):
        data_dict = self.rnn_encoding(data_dict["lang_feat"], data_dict["lang_len"], data_dict)
        if self.use_lang_classifier:
            data_dict["lang_scores"] = self.lang_cls(data_dict["lang_cls_feats"])
        return data_dict
