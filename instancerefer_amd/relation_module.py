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
from .data import idx_tensor, selection_on_device, upload_instances
from .dense import cosine_rows, mlp2
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

    def node_features(self, data_dict, cls_list):
        """Everything of this module that depends only on the inputs: candidate selection, its device index tensors
        and the per-instance node features [box centre, mean colour/height, one-hot class]. -> (sel, sd, centres, feats)
        or None when no scene has >= 2 candidates. Parameter-free, so a training loop can run it in its
        input-preparation stage (InstanceRefer.prepare does when use_gt_lang)."""
        pack = upload_instances(data_dict)
        sel = pack.select(cls_list)
        if len(sel['cand']) == 0:
            return None
        dev = pack.pts32.device
        sd = selection_on_device(sel, pack, dev)
        support = sd['support']
        mean = F_.segment_mean(pack.pts32.index_select(0, support))          # (S, C0)
        centres = pack.centres.index_select(0, support)                     # (S, 3)
        mean = torch.cat([centres, mean[:, 3:]], 1)                         # xyz <- box centre
        onehot = nn.functional.one_hot(sd['support_class'], self.args.num_classes).to(mean.dtype)
        feats = torch.cat([mean, onehot], 1)                                # (S, C0 + num_classes)
        # the kNN grid of the edge convolution depends on the boxes only: built here (input preparation), not in the head
        qis = sd['query_in_support']
        nbr = F_.knn_batched(centres, sd['support_offsets'], torch.index_select(centres, 0, qis),
                             torch.index_select(sd['support_seg'], 0, qis).int(), self.gcn.k)
        return sel, sd, centres, feats, nbr

    def prepare(self, data_dict, cls_list):
        data_dict['_rel_prepared'] = (self.node_features(data_dict, cls_list),)
        return data_dict

    def forward(self, data_dict):
        lang_feats_in = data_dict['lang_rel_feats']
        if '_rel_prepared' in data_dict:
            prep = data_dict.pop('_rel_prepared')[0]
        else:
            cls_list = data_dict.get('_lang_cls_pred_list')
            if cls_list is None:
                lang_cls_pred = (data_dict['object_cat'] if self.args.use_gt_lang
                                 else torch.argmax(data_dict["lang_scores"], dim=1))
                cls_list = lang_cls_pred.tolist()
            prep = self.node_features(data_dict, cls_list)
        if prep is None:
            mlp2(self.lang_emb_fc, lang_feats_in)                            # (the reference runs the language MLP before it looks)
            data_dict['relation_scores'] = lang_feats_in.new_zeros((0,))
            return data_dict
        from . import heads
        if heads.relation_head(self, lang_feats_in, prep, data_dict):       # the whole head as ONE autograd node (csrc/heads_nodes.cpp)
            return data_dict
        lang_feats = mlp2(self.lang_emb_fc, lang_feats_in)                   # (B, h_dim)
        sel, sd, centres, feats = prep[:4]
        # batch ids of the support rows are renumbered over the kept scenes (contiguous segments)
        feats = self.gcn(centres, sd['support_seg'], sd['query_in_support'], feats, support_offsets=sd['support_offsets'],
                         nbr=prep[4] if len(prep) > 4 else None)
        feats = mlp2(self.vis_emb_fc, feats)
        data_dict['relation_scores'] = cosine_rows(feats, lang_feats, sd['cand_scene'])      # F.cosine_similarity, eps 1e-8
        return data_dict
