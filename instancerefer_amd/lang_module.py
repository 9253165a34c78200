"""LangModule — drop-in for the reference's models/lang_module.py:7-108 (same ctor signature, state-dict
keys `gru.* word_projection.* fc_a fc_cls fc_rel fc_scene lang_cls.0`, and data_dict keys).

GloVe(300) -> MLP 300->256->256 -> 2-layer (bi)GRU(128) -> four attention heads that pool the
*MLP-projected* embeddings (softmax over padded positions, then mask + renormalise; reference
lang_module.py:61-83) -> 4 x 256-d sentence vectors + 18-way classifier. Dense work: PyTorch-ROCm
(own persistent GRU recurrence kernel on HIP devices, GEMMs on MFMA); the four heads are evaluated as one batched GEMM.

Training on a HIP device replays the whole module — ~70 forward and ~110 backward launches of microsecond kernels on
static shapes (B, max(len)) — from a hipGraph captured once per shape (`torch.cuda.make_graphed_callables`): ~0.6 ms of
host dispatch per step becomes two graph launches (the step is host-bound in the bf16 modes). Same kernels, same
order: bit-identical to the eager module (tests/test_model_gpu.py). IRX_LANG_GRAPH=0 switches it off.
"""
import os
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from .dense import gru_packed


class _LangCore(nn.Module):
    """The tensor-in / tensor-out body of LangModule.forward that a hipGraph can replay."""

    def __init__(self, lang, t_max):
        super().__init__()
        self.lang, self.t_max = lang, t_max

    def forward(self, feat, length):
        dd = self.lang._eager({"lang_feat": feat, "lang_len": length, "lang_len_max": self.t_max})
        return (dd["lang_feat"], dd["_att"], dd["_pooled"], dd["lang_scores"])


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
        mask = (torch.arange(t_max, device=feats.device).unsqueeze(0) <
                length.to(feats.device).unsqueeze(1)).to(feats.dtype)
        # the four heads as one (o_dim x 4) projection; order: attr, cls, rel, scene
        w = torch.cat([self.fc_a.weight, self.fc_cls.weight, self.fc_rel.weight, self.fc_scene.weight], 0)
        b = torch.cat([self.fc_a.bias, self.fc_cls.bias, self.fc_rel.bias, self.fc_scene.bias], 0)
        att = torch.softmax(feats.matmul(w.t()) + b, dim=1)       # softmax over ALL T_max positions
        att = att * mask.unsqueeze(2)
        att = att / att.sum(1, keepdim=True)                      # (B, T_max, 4)
        pooled = torch.bmm(att.transpose(1, 2), embed[:, :t_max])  # (B, 4, 256)
        data_dict['_att'], data_dict['_pooled'] = att, pooled
        return self._publish(data_dict, att, pooled)

    @staticmethod
    def _publish(data_dict, att, pooled):
        data_dict['atten_attr'] = att[:, :, 0]
        data_dict['atten_rel'] = att[:, :, 2]
        data_dict['atten_scene'] = att[:, :, 3]
        data_dict['lang_attr_feats'] = pooled[:, 0]
        data_dict['lang_cls_feats'] = pooled[:, 1]
        data_dict['lang_rel_feats'] = pooled[:, 2]
        data_dict['lang_scene_feats'] = pooled[:, 3]
        return data_dict

    def _eager(self, data_dict):
        data_dict = self.rnn_encoding(data_dict["lang_feat"], data_dict["lang_len"], data_dict)
        if self.use_lang_classifier:
            data_dict["lang_scores"] = self.lang_cls(data_dict["lang_cls_feats"])
        return data_dict

    def _graphed(self, feat, length, t_max):
        """The captured module for this (batch, max(len)) shape, or None when capture is not possible. The cache is
        invalidated when the parameters move (optim.FlatAdam re-homes them into its flat buffer once)."""
        cache = self.__dict__.setdefault('_graphs', {})
        anchor = tuple(p.data_ptr() for p in self.parameters())
        if cache.get('anchor') != anchor:
            cache.clear()
            cache['anchor'] = anchor
        key = (tuple(feat.shape), t_max, torch.cuda.current_device())
        g = cache.get(key)
        if g is None:
            if len(cache) > 40:          # one graph per distinct max(len): bounded
                return None
            try:
                core = _LangCore(self, t_max)
                g = torch.cuda.make_graphed_callables(core, (feat.detach().clone(), length.detach().clone()))
            except Exception as e:       # capture unsupported in this context: stay eager (and say so once)
                import warnings
                warnings.warn("LangModule: hipGraph capture failed (%r); running eagerly" % (e,))
                g = False
            cache[key] = g
        return g or None

    def forward(self, data_dict):
        feat, length = data_dict["lang_feat"], data_dict["lang_len"]
        if (feat.is_cuda and self.training and torch.is_grad_enabled() and self.use_lang_classifier
                and "lang_len_max" in data_dict and not feat.requires_grad
                and os.environ.get("IRX_LANG_GRAPH", "1") != "0" and not torch.cuda.is_current_stream_capturing()):
            t_max = int(data_dict["lang_len_max"])
            g = self._graphed(feat[:, :t_max], length, t_max)
            if g is not None:
                feats, att, pooled, scores = g(feat[:, :t_max].contiguous(), length)
                data_dict['lang_feat'] = feats
                data_dict["lang_scores"] = scores
                return self._publish(data_dict, att, pooled)
        data_dict = self._eager(data_dict)
        data_dict.pop('_att', None)
        data_dict.pop('_pooled', None)
        return data_dict
