"""AttributeModule — drop-in for the reference's models/attribute_module.py:11-131.

Per same-class candidate instance: voxelise (2 cm) -> SparseConvEncoder -> global max-pool -> MLP ->
cosine score against the language attribute vector. The reference voxelises each candidate with numpy
inside forward (a host double loop, attribute_module.py:59-71) and uploads the collated voxels every
step; here all candidates of the batch are voxelised by ONE hash-voxelisation pass on the GPU
(float64 floor(x / v), first-occurrence representative, batch index = candidate index).
"""
import numpy as np
import torch
import torch.nn as nn

from .basic_blocks import SparseConvEncoder
from .data import idx_tensor, selection_on_device, upload_instances
from .sparse.encoder_fn import lane_of, lane_wait
from .dense import cosine_rows, mlp2
from .sparse import nn as spnn
from .sparse.utils import voxelize, voxelize_launch


class AttributeModule(nn.Module):
    def __init__(self, input_feature_dim, args, v_dim=128, h_dim=256, l_dim=256):
        super().__init__()
        self.args = args
        self.input_feature_dim = input_feature_dim
        self.voxel_size = np.array([args.voxel_size_ap] * 3)
        self.net = SparseConvEncoder(self.input_feature_dim)
        self.pooling = spnn.GlobalMaxPooling()
        self.vis_emb_fc = nn.Sequential(nn.Linear(v_dim, h_dim), nn.LayerNorm(h_dim), nn.ReLU(),
                                        nn.Linear(h_dim, h_dim))
        self.lang_emb_fc = nn.Sequential(nn.Linear(l_dim, h_dim), nn.BatchNorm1d(h_dim), nn.ReLU(),
                                         nn.Linear(h_dim, h_dim))
        self.weight_initialization()

    def weight_initialization(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def filter_candidates(self, data_dict, lang_cls_pred, launch_only=False):
        """-> (SparseTensor of all candidates' voxels, pred_obb_batch, num_filtered_objs); same selection
        rule as the reference (class match; scenes with < 2 candidates contribute no voxels).
        launch_only: -> (VoxelizePending, sel): voxeliser + 4-level pyramid enqueued, no host sync yet."""
        pack = upload_instances(data_dict)
        sel = pack.select(lang_cls_pred)
        dev = pack.pts32.device
        nc = len(sel['cand'])
        if nc == 0:
            return None, sel
        cand = selection_on_device(sel, pack, dev)['cand']
        xyz = pack.xyz64.index_select(0, cand)                 # (Nc, P, 3) float64
        pts = pack.pts32.index_select(0, cand)                 # (Nc, P, C0) float32
        p = xyz.shape[1]
        batch = torch.arange(nc, device=dev, dtype=torch.int32).repeat_interleave(p)
        if launch_only:
            return voxelize_launch(xyz.view(-1, 3), pts.view(nc * p, -1), batch, self.voxel_size, nc, 4), sel
        st = voxelize(xyz.view(-1, 3), pts.view(nc * p, -1), batch, self.voxel_size, nc)
        return st, sel

    def prepare(self, data_dict, lang_cls_pred):
        """Everything of this module that needs a host sync (class list -> candidate selection -> voxel count ->
        pyramid level sizes) and depends only on the inputs. InstanceRefer.forward runs it BEFORE queueing the heavy
        GPU work (when use_gt_lang), so these syncs never wait behind a long queue."""
        return self.prepare_finish(self.prepare_launch(data_dict, lang_cls_pred))

    def prepare_launch(self, data_dict, lang_cls_pred):
        """prepare() with every kernel enqueued and NO host sync: the voxel count and the pyramid level sizes are
        collected by prepare_finish()."""
        pending, sel = self.filter_candidates(data_dict, lang_cls_pred, launch_only=True)
        data_dict['_attr_pending'] = pending
        data_dict['_attr_prepared'] = (None, sel)
        data_dict['_lang_cls_pred_list'] = list(lang_cls_pred)
        return data_dict

    def prepare_finish(self, data_dict):
        pending = data_dict.pop('_attr_pending', None)
        if pending is not None:
            data_dict['_attr_prepared'] = (pending.finish(), data_dict['_attr_prepared'][1])
        return data_dict

    def encode(self, data_dict, defer=False):
        """Issue the candidate encoder now if the candidates are already known (prepare() ran: GT classes) — it needs
        nothing from the language module. forward() picks the result up; identical to running it there.
        defer=True: the pass is issued now, its autograd node is created when the caller attaches it (encoder_fn.Deferred) — the
        backward reaches the encoder in the order of the attach."""
        prep = data_dict.get('_attr_prepared')
        if prep is not None and prep[0] is not None and '_attr_encoded' not in data_dict:
            data_dict['_attr_encoded'] = self.net(prep[0], defer=defer)
        return data_dict

    def forward(self, data_dict):
        lang_feats = data_dict.pop('_attr_lang_h', None)                      # already through lang_emb_fc (heads.PreLang) ...
        if lang_feats is None:
            lang_feats = mlp2(self.lang_emb_fc, data_dict['lang_attr_feats'])  # ... or here: (B, h_dim)

        if '_attr_prepared' in data_dict:
            st, sel = data_dict.pop('_attr_prepared')
        else:
            if not self.args.use_gt_lang:
                lang_cls_pred = torch.argmax(data_dict["lang_scores"], dim=1)
            else:
                lang_cls_pred = data_dict['object_cat']
            lang_cls_pred = lang_cls_pred.tolist()   # one D2H of B ints (the reference syncs per instance)
            data_dict['_lang_cls_pred_list'] = lang_cls_pred      # reused by the relation module (no second sync)
            st, sel = self.filter_candidates(data_dict, lang_cls_pred)
        data_dict['num_filtered_objs'] = sel['num_filtered_objs']
        data_dict['pred_obb_batch'] = sel['pred_obb_batch']
        dev = lang_feats.device
        if st is None:   # no scene with >= 2 candidates (the reference crashes here: torch.cat([]))
            data_dict['obj_feats'] = lang_feats.new_zeros((0, 128))
            data_dict['attribute_scores'] = lang_feats.new_zeros((0,))
            return data_dict

        feats = data_dict.pop('_attr_encoded', None)
        if feats is None:
            feats = self.net(st)
        elif hasattr(feats, 'attach'):
            feats = feats.attach()
        lane_wait(lane_of(self.net))                      # the encoder may be issued by a library thread
        feats = self.pooling(feats)                       # (Nc, 128)
        data_dict['obj_feats'] = feats
        feats = mlp2(self.vis_emb_fc, feats)
        sd = selection_on_device(sel, upload_instances(data_dict), dev)
        data_dict['_sel_dev'] = sd                         # the scene head reuses cand_scene
        # normalize(vis) . normalize(lang)[scene of the candidate]  ==  a clamped cosine (F.normalize eps = 1e-12)
        data_dict['attribute_scores'] = cosine_rows(feats, lang_feats, sd['cand_scene'], eps=1e-12)
        return data_dict
