"""SceneModule — drop-in for the reference's models/scene_module.py:9-108.

Whole-scene BEVEncoder (5 cm voxels) -> crop [0,240)x[0,400)x[0,80) -> dense BEV (15x25, per-z 128x128
kernels summed over 5 z-bins) -> BN2d/ReLU -> 2 x Conv2d 3x3 -> language-guided attention over the 11x21
cells -> 9-way area classifier + cosine score between the scene vector and each candidate's obj_feats.
"""
import math

import torch
import torch.nn as nn

from .basic_blocks import BEVEncoder, SparseCrop, ToDenseBEVConvolution, batchnorm_rows, conv2d_rows
from .data import idx_tensor
from .dense import AttentionPoolFn, cosine_rows, mlp2
from .sparse.encoder_fn import lane_of, lane_wait
from .sparse import nn as spnn


import os as _os
_DEFER = _os.environ.get('IRX_SCENE_DEFER', '1') != '0'     # dev A/B switch (bit-identical results)
_FUSED_ATTN = _os.environ.get('IRX_FUSED_ATTN', '1') != '0'  # dense.AttentionPoolFn (one launch each way) instead of bmm / softmax / mul / sum through ATen


class SceneModule(nn.Module):
    def __init__(self, input_feature_dim, args, v_dim=128, h_dim=128, l_dim=256, dropout_rate=0.15):
        super().__init__()
        self.args = args
        self.input_feature_dim = input_feature_dim
        self.net = BEVEncoder(self.input_feature_dim)
        self.pooling = spnn.GlobalMaxPooling()   # constructed but unused, as in the reference (:20)
        loc_max = [240, 400, 80]
        loc_min = [0, 0, 0]
        shape = [(a - b) // 16 for a, b in zip(loc_max, loc_min)]
        self.to_bev = nn.Sequential(
            SparseCrop(loc_min=loc_min, loc_max=loc_max),
            ToDenseBEVConvolution(128, 128, shape=shape, z_dim=2, offset=loc_min),
            nn.BatchNorm2d(128),
            nn.ReLU(True),
        )
        self.h_dim = h_dim
        self.vis_emb_fc = nn.Sequential(nn.Conv2d(v_dim, h_dim, 3), nn.BatchNorm2d(h_dim), nn.ReLU(),
                                        nn.Dropout(dropout_rate), nn.Conv2d(h_dim, h_dim, 3))
        self.vis_emb_fc1 = nn.Sequential(nn.Linear(128, h_dim), nn.LayerNorm(h_dim), nn.ReLU(),
                                         nn.Dropout(dropout_rate), nn.Linear(h_dim, h_dim))
        self.lang_emb_fc = nn.Sequential(nn.Linear(l_dim, h_dim), nn.LayerNorm(h_dim), nn.ReLU(),
                                         nn.Dropout(dropout_rate), nn.Linear(h_dim, h_dim))
        self.cls = nn.Sequential(nn.Linear(h_dim, h_dim), nn.BatchNorm1d(h_dim), nn.ReLU(), nn.Linear(h_dim, 9))

    def encode(self, data_dict):
        """Run the whole-scene BEVEncoder now (it depends on `lidar` only). InstanceRefer.forward calls this before the
        language / attribute / relation modules so that this ~5 ms of GPU work overlaps their host-side work; the
        result is identical to running it inside forward()."""
        feats = data_dict['lidar']
        if feats._batch_size is None:
            feats._batch_size = data_dict['point_min'].shape[0]
        # deferred node: issued now, attached at the head of forward() so that the backward replays it right behind this module's
        # head instead of last (sparse/encoder_fn.py: Launched)
        data_dict['_scene_encoded'] = self.net(feats, defer=_DEFER)
        return data_dict

    def head(self, data_dict):
        """Everything of forward() that does not need the candidates' features: encoder output -> BEV -> BatchNorm / ReLU ->
        2 x Conv2d -> language attention -> scene vector + area classifier (reference scene_module.py:61-96). InstanceRefer's
        multi-stream forward issues it on the scene encoder's own stream, right behind the encoder and beside the candidate
        encoder; forward() runs it when nobody has."""
        feats = data_dict['lidar']
        batch_size = data_dict['point_min'].shape[0]
        lang_feats = data_dict['lang_scene_feats']
        if '_scene_encoded' in data_dict:
            feats = data_dict.pop('_scene_encoded')
            if hasattr(feats, 'attach'):
                feats = feats.attach()
        else:
            if feats._batch_size is None:
                feats._batch_size = batch_size   # known from the collate; avoids the reference's .item() sync
            feats = self.net(feats)
        lane_wait(lane_of(self.net))             # the encoder may be issued by a library thread (encoder_fn.py)
        from . import heads
        if heads.scene_head(self, feats, data_dict):     # everything below as ONE autograd node (csrc/heads_nodes.cpp)
            return data_dict
        # SparseCrop (to_bev[0]) is folded into the BEV gather: only voxels inside the window are looked up.
        # The dense head runs on channels-last cell rows (cells, C) with the irx conv / BatchNorm kernels.
        nx, ny = self.to_bev[1].bev_shape
        rows, _ = self.to_bev[1].rows(feats)                                    # (B*15*25, 128)
        rows = batchnorm_rows(self.to_bev[2], rows, relu=True)                  # BatchNorm2d + ReLU
        rows = conv2d_rows(self.vis_emb_fc[0], rows, batch_size, nx, ny)        # Conv2d 3x3 -> (B*13*23, D)
        rows = batchnorm_rows(self.vis_emb_fc[1], rows, relu=True)
        rows = self.vis_emb_fc[3](rows)                                         # Dropout
        rows = conv2d_rows(self.vis_emb_fc[4], rows, batch_size, nx - 2, ny - 2)  # -> (B*11*21, D)
        h, w = nx - 4, ny - 4
        feats = rows.view(batch_size, h * w, self.h_dim)                        # (B, n_vis, D)
        pre = data_dict.pop('_scene_lang_h', None)                              # already through lang_emb_fc (heads.PreLang)
        lang_feats = pre if pre is not None else mlp2(self.lang_emb_fc, lang_feats)
        if feats.is_cuda and _FUSED_ATTN and feats.shape[2] <= 256:
            atten, scene_feats = AttentionPoolFn.apply(feats, lang_feats)     # one launch each way (csrc/irx_match.hip)
        else:
            atten = torch.bmm(feats, lang_feats.unsqueeze(2)) / math.sqrt(feats.shape[2])
            atten = torch.softmax(atten.squeeze(2), dim=1)
            scene_feats = torch.sum(feats * atten.unsqueeze(2), dim=1)
        data_dict['vis_atten'] = atten.reshape(batch_size, h, w)

        data_dict['seg_scores'] = mlp2(self.cls, scene_feats)
        data_dict['_scene_feats'] = scene_feats
        return data_dict

    def forward(self, data_dict):
        if '_scene_feats' not in data_dict:
            data_dict = self.head(data_dict)
        scene_feats = data_dict.pop('_scene_feats')
        batch_size = data_dict['point_min'].shape[0]
        pred_obb_batch = data_dict['pred_obb_batch']
        obj_feats_flatten = data_dict['obj_feats']
        cand_scene = [i for i in range(batch_size) for _ in range(len(pred_obb_batch[i]))
                      if len(pred_obb_batch[i]) >= 2]
        if len(cand_scene) == 0:
            data_dict['scene_scores'] = scene_feats.new_zeros((0,))
            return data_dict
        sd = data_dict.get('_sel_dev')
        cs = sd['cand_scene'] if sd is not None else idx_tensor(cand_scene, scene_feats.device)
        obj = mlp2(self.vis_emb_fc1, obj_feats_flatten)
        data_dict['scene_scores'] = cosine_rows(obj, scene_feats, cs)                        # F.cosine_similarity, eps 1e-8
        return data_dict
