"""RelationModule — drop-in for the reference's models/relation_module.py:7-107.

Instance graph: node feature = [mean of the instance's 1024 points with xyz <- box centre, one-hot class];
queries = same-class candidates, support = all instances of scenes with >= 2 candidates; kNN (k=8,
self included) edge-conv with max aggregation -> MLP -> cosine score vs the language relation vector.
The reference computes the means in a numpy loop and uploads four tensors per step
(relation_module.py:59-76,94-98); here the means are one segmented-mean launch over the resident pack.
"""
import numpy as np
import torch
import torch.nn as nn

from .basic_blocks import DynamicEdgeConv
from .data import idx_tensor, upload_instances
from .sparse import functional as F_


class RelationModule(nn.Module):
    def __init__(self, input_feature_dim, args, v_dim=128, h_dim=128, l_dim=256, dropout_rate=0.15):
        super().__init__()
        self.args = args
        self.input_feature_dim = input_feature_dim
        self.vis_emb_fc = nn.Sequential(nn.Linear(v_dim, h_dim), nn.LayerNorm(h_dim), nn.ReLU(),
                                        nn.Dropout(dropout_rate), nn.Linear(h_dim, h_dim))
        self.lang_emb_fc = nn.Sequential(nn.Linear(l_dim, h_dim), nn.BatchNorm1d(h_dim), nn.ReLU(),
                                         nn.Dropout(dropout_rate), nn.Linear(h_dim, h_dim))
        self.gcn = DynamicEdgeConv(input_feature_dim + args.num_classes, 128, k=args.k,
                                   num_classes=args.num_classes)
        self.one_hot_array = np.eye(args.num_classes)
        self.weight_initialization()

    def weight_initialization(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, data_dict):
        lang_feats = self.lang_emb_fc(data_dict['lang_rel_feats'])           # (B, h_dim)
        if not self.args.use_gt_lang:
            lang_cls_pred = torch.argmax(data_dict["lang_scores"], dim=1)
        else:
            lang_cls_pred = data_dict['object_cat']
        pack = upload_instances(data_dict)
        cls_list = data_dict.get('_lang_cls_pred_list')
        sel = pack.select(cls_list if cls_list is not None else lang_cls_pred.tolist())
        dev = lang_feats.device
        if len(sel['cand']) == 0:
            data_dict['relation_scores'] = lang_feats.new_zeros((0,))
            return data_dict

        support = idx_tensor(sel['support'], dev)
        mean = F_.segment_mean(pack.pts32.index_select(0, support))          # (S, C0)
        centres = pack.centres.index_select(0, support)                     # (S, 3)
        mean = torch.cat([centres, mean[:, 3:]], 1)                         # xyz <- box centre
        onehot = nn.functional.one_hot(idx_tensor([pack.classes[s] for s in sel['support']], dev),
                                       self.args.num_classes).to(mean.dtype)
        feats = torch.cat([mean, onehot], 1)                                # (S, C0 + num_classes)

        # batch ids of the support rows, renumbered over the kept scenes (contiguous segments)
        kept = sel['support_scene_offsets']
        seg_of = np.repeat(np.arange(len(kept) - 1), np.diff(kept))
        batch_index = idx_tensor(seg_of, dev)
        filtered_index = idx_tensor(sel['query_in_support'], dev)
        sup_off = idx_tensor(kept, dev, torch.int32)

        feats = self.gcn(centres, batch_index, filtered_index, feats, support_offsets=sup_off)
        feats = self.vis_emb_fc(feats)
        lang_flat = lang_feats.index_select(0, idx_tensor(sel['cand_scene'], dev))
        data_dict['relation_scores'] = nn.functional.cosine_similarity(feats, lang_flat, dim=1)
        return data_dict
