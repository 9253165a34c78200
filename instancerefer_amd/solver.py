"""Solver — DDP-aware sibling of the reference's lib/solver.py:139-342.

Same contract: iterate dataloaders that yield reference-format `data_dict`s (lib/dataset.py collate), move the tensor
keys to the GPU (lib/solver.py:242-245), forward -> get_loss -> backward -> step, get_eval for the metrics, log every
`verbose` iterations, save `model_last.pth` every epoch, `model.pth` on the best Acc@0.25 and `checkpoint.tar`
({epoch, model_state_dict, optimizer_state_dict}) at the end — the reference's file names and state-dict keys;
`optimizer_state_dict` is in torch.optim.Adam.state_dict() layout (optim.FlatAdam.state_dict), so the reference's
`optimizer.load_state_dict(checkpoint["optimizer_state_dict"])` (scripts/train.py:114-119) reads it and
`Solver(use_checkpoint=...)` / `Solver.load_checkpoint` resumes from either side's file. The scalars the reference
sends to tensorboardX (lib/solver.py:344-366: loss/{loss,ref_loss,lang_loss,seg_loss}, score/{lang_acc,ref_acc,seg_acc,
iou_rate_0.25,iou_rate_0.5}) go to `scalars.jsonl` under the same tags (and to a SummaryWriter when tensorboard is
importable; it is not in this image).

Differences, all below the API: one process per GPU with a single flat-gradient RCCL all-reduce and one fused Adam
launch per step (optim.FlatAdam); rank 0 alone logs and writes files; timers are taken around device syncs only at
logging points (the reference's `time.time()` pairs measure host time without a sync)."""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .eval_helper import get_eval
from .loss_helper import get_loss
from .optim import FlatAdam

GPU_KEYS = ("lang_feat", "lang_len", "object_cat", "lidar", "point_min", "point_max", "ref_center_label",
            "ref_size_residual_label")
HOST_LABELS = ("ref_center_label", "ref_size_residual_label", "ref_heading_class_label",
               "ref_heading_residual_label", "ref_size_class_label", "object_cat", "unique_multiple", "point_min",
               "point_max")


def to_device(data_dict, device):
    """lib/solver.py:242-245 + host copies of the label tensors (they originate on the host) for get_loss / get_eval."""
    host = dict(data_dict.get("_host") or {})      # a device-side loader (scene_input.ResidentLoader) brings its own
    for k in HOST_LABELS:
        if k not in host and isinstance(data_dict.get(k), torch.Tensor):
            host[k] = data_dict[k].detach().cpu().numpy()
    data_dict["_host"] = host
    if "lang_len" in data_dict:
        data_dict["lang_len_max"] = int(data_dict["lang_len"].max())
    for k in GPU_KEYS:
        v = data_dict.get(k)
        if isinstance(v, torch.Tensor):
            data_dict[k] = v.to(device, non_blocking=True)
        elif v is not None and hasattr(v, "cuda") and not isinstance(v, (list, tuple)):
            data_dict[k] = v.to(device)          # SparseTensor
    return data_dict


def scheduled_lr(base_lr, epoch, lr_decay_step, lr_decay_rate):
    """MultiStepLR (reference lib/solver.py:119-124): base_lr * rate ** (number of milestones <= epoch)."""
    return base_lr * (lr_decay_rate ** sum(epoch >= s for s in lr_decay_step))


def resume_base_lr(saved_lr, initial_lr, start_epoch, lr_decay_step, lr_decay_rate):
    """The UNDECAYED rate of a resumed run. `saved_lr` is the group's lr in the checkpoint — already decayed by the
    milestones the interrupted run had passed — so using it as the base would decay it a second time. `initial_lr` (what
    torch's schedulers and this Solver record in the param group) wins; without it the decay of the last epoch trained
    (start_epoch - 1) is divided out."""
    if initial_lr is not None:
        return float(initial_lr)
    last = max(int(start_epoch) - 1, 0)
    return float(saved_lr) / (lr_decay_rate ** sum(last >= s for s in lr_decay_step))


class Solver:
    def __init__(self, model, config, dataloader, lr=1e-3, weight_decay=1e-5, lr_decay_step=(15, 20), lr_decay_rate=0.1,
                 out_dir=None, verbose=20, device=None, use_checkpoint=None, sync_bn=False):
        if not torch.cuda.is_initialized():        # the queue count is read once, when the HIP runtime starts (configure_hw_queues)
            from . import configure_hw_queues
            configure_hw_queues()
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.model = model.to(self.device)
        self.sync_bn = bool(sync_bn)
        self.sync_bn_skipped = 0                   # steps dropped by sync_bn_guard (same count on every rank)
        if sync_bn:                                # world > 1: BatchNorm statistics over all ranks (syncbn.py)
            from .syncbn import convert_sync_batchnorm
            convert_sync_batchnorm(self.model)
        self.config = config
        self.dataloader = dataloader               # {"train": iterable, "val": iterable (optional)}
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        # world > 1: rank 0's parameters and buffers are broadcast, so per-rank seeds cannot make the replicas diverge
        self.optimizer = FlatAdam(self.model.parameters(), lr=lr, weight_decay=weight_decay, world_size=self.world,
                                  module=self.model)
        self.start_epoch = 0
        self.base_lr, self.lr_decay_step, self.lr_decay_rate = lr, tuple(lr_decay_step or ()), lr_decay_rate
        self.optimizer.initial_lr = lr
        self.out_dir = out_dir
        self.verbose = verbose
        self.best = {"epoch": 0, "iou_rate_0.25": -float("inf"), "iou_rate_0.5": -float("inf"), "ref_acc": -float("inf")}
        self.log = {"train": [], "val": []}
        self.global_iter = 0
        if self.rank == 0 and out_dir:
            os.makedirs(out_dir, exist_ok=True)
        self._writer = None
        if self.rank == 0 and out_dir:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self._writer = SummaryWriter(os.path.join(out_dir, "tensorboard"))
            except Exception:                      # tensorboard is absent in this image: scalars.jsonl only
                self._writer = None
        if use_checkpoint:
            self.load_checkpoint(use_checkpoint)

    # ------------------------------------------------------------------------------------------
    def _say(self, msg):
        if self.rank == 0:
            print(msg, flush=True)
            if self.out_dir:
                with open(os.path.join(self.out_dir, "log.txt"), "a") as f:
                    f.write(msg + "\n")

    def _scalars(self, phase, rec):
        """lib/solver.py:344-366 (_dump_log): the same tags, one JSON line per logging point."""
        if self.rank != 0 or not self.out_dir:
            return
        tags = {}
        for k in ("loss", "ref_loss", "lang_loss", "seg_loss"):
            if k in rec:
                tags["loss/" + k] = rec[k]
        for k, src in (("lang_acc", "lang_acc"), ("ref_acc", "ref_acc"), ("seg_acc", "seg_acc"),
                       ("iou_rate_0.25", "iou_rate_25"), ("iou_rate_0.5", "iou_rate_5")):
            v = rec.get(src, rec.get(k))
            if v is not None:
                tags["score/" + k] = float(v)
        import json
        with open(os.path.join(self.out_dir, "scalars.jsonl"), "a") as f:
            f.write(json.dumps({"phase": phase, "iter": self.global_iter, **tags}) + "\n")
        if self._writer is not None:
            for t, v in tags.items():
                self._writer.add_scalar("%s/%s" % (phase, t), v, self.global_iter)

    def load_checkpoint(self, path):
        """Resume (reference scripts/train.py:114-119): checkpoint.tar written by this Solver OR by the reference's."""
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(ck["model_state_dict"])      # in place: parameters stay views of the flat buffer
        self.optimizer.load_state_dict(ck["optimizer_state_dict"])
        self.start_epoch = int(ck.get("epoch", 0))
        self.base_lr = resume_base_lr(self.optimizer.lr, self.optimizer.initial_lr, self.start_epoch, self.lr_decay_step,
                                      self.lr_decay_rate)
        self.optimizer.initial_lr = self.base_lr
        return ck

    def _forward(self, data_dict):
        return get_loss(self.model(to_device(data_dict, self.device)), self.config)

    def has_scored_candidates(self, data_dict):
        """Host-side: does this rank's shard hold a scene with >= 2 instances of the target class (the condition under which
        the candidate encoder and the attribute / relation heads run at all, reference models/attribute_module.py:75-76)?
        Known before the forward only when the target class is the ground-truth one (`use_gt_lang`, the training default)."""
        args = getattr(self.model, "args", None)
        if args is not None and not getattr(args, "use_gt_lang", True):
            return True                            # arg-max of lang_scores: not known on the host before the forward
        host = data_dict.get("_host") or {}
        src = host["object_cat"] if "object_cat" in host else data_dict["object_cat"]      # (the host copy: no D2H sync)
        cats = [int(c) for c in torch.as_tensor(src).reshape(-1).tolist()]
        return any(sum(int(c) == cat for c in cls) >= 2 for cls, cat in zip(data_dict["instance_class"], cats))

    def sync_bn_guard(self, data_dict):
        """Sync-BatchNorm contract (syncbn.py): every rank runs every BatchNorm layer in every step. A shard without scored
        candidates would skip the candidate encoder and the heads' BatchNorm1d layers and leave the other ranks waiting in
        their collectives. One 4-byte all-reduce (MIN) before the forward makes the decision the same everywhere: the step
        runs only if EVERY rank can run every layer, otherwise all ranks drop this batch (counted in sync_bn_skipped)."""
        if not (self.sync_bn and self.world > 1):
            return True
        args = getattr(self.model, "args", None)
        if args is not None and not getattr(args, "use_gt_lang", True):
            # the candidate set then depends on arg-max(lang_scores), unknown before the forward: the guard cannot promise that every
            # rank runs every layer, and a rank without candidates would leave the others hanging in a collective
            raise RuntimeError("sync BatchNorm with use_gt_lang=False is not supported: whether a rank runs the candidate encoder "
                               "is only known inside the forward; train with the per-rank BatchNorm or with use_gt_lang=True")
        flag = torch.tensor([1 if self.has_scored_candidates(data_dict) else 0], dtype=torch.int32,
                            device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item()) == 1
        if not ok:
            self.sync_bn_skipped += 1
            if self.sync_bn_skipped in (1, 10, 100) or self.sync_bn_skipped % 1000 == 0:
                self._say("sync-BN: batch dropped on all ranks (a shard has no scene with >= 2 candidates); %d so far"
                          % self.sync_bn_skipped)
        return ok

    def train_epoch(self, epoch):
        self.model.train()
        self.optimizer.lr = scheduled_lr(self.base_lr, epoch, self.lr_decay_step, self.lr_decay_rate)
        t0, seen = time.perf_counter(), 0
        for data_dict in self.dataloader["train"]:
            if not self.sync_bn_guard(data_dict):
                continue
            self.optimizer.zero_grad()
            data_dict = self._forward(data_dict)
            loss = data_dict["loss"]
            one = self.__dict__.get("_unit_grad")     # (saves the ones_like allocation + fill launch of a bare backward() per step)
            if one is None or one.device != loss.device or one.shape != loss.shape or one.dtype != loss.dtype:
                one = self.__dict__["_unit_grad"] = torch.ones_like(loss)
            loss.backward(gradient=one)
            self.optimizer.backward_step()
            self.global_iter += 1
            seen += data_dict["lang_scores"].shape[0]
            if self.global_iter % self.verbose == 0:
                with torch.no_grad():
                    ev = get_eval(data_dict, self.config)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                rec = dict(epoch=epoch, iter=self.global_iter, loss=float(data_dict["loss"].detach()),
                           ref_loss=float(data_dict["ref_loss"].detach()),
                           lang_loss=float(data_dict["lang_loss"].detach()), seg_loss=float(data_dict["seg_loss"].detach()),
                           lang_acc=float(ev["lang_acc"]), ref_acc=float(np.mean(ev["ref_acc"])),
                           iou_rate_25=ev["ref_iou_rate_0.25"], iou_rate_5=ev["ref_iou_rate_0.5"],
                           scenes_per_sec=self.world * seen / dt)
                rec["seg_acc"] = float(data_dict["seg_acc"].detach()) if "seg_acc" in data_dict else None
                self.log["train"].append(rec)
                self._scalars("train", rec)
                self._say("[train] epoch %d iter %d loss %.4f (ref %.4f lang %.4f seg %.4f) ref_acc %.3f Acc@.25 %.3f "
                          "%.1f scenes/s" % (epoch, rec["iter"], rec["loss"], rec["ref_loss"], rec["lang_loss"],
                                             rec["seg_loss"], rec["ref_acc"], rec["iou_rate_25"], rec["scenes_per_sec"]))
                t0, seen = time.perf_counter(), 0

    @torch.no_grad()
    def validate(self, epoch):
        if not self.dataloader.get("val"):
            return None
        self.model.eval()
        acc, ious = [], []
        for data_dict in self.dataloader["val"]:
            ev = get_eval(self._forward(data_dict), self.config)
            acc += ev["ref_acc"]
            ious += ev["ref_iou"]
        stats = torch.tensor([np.sum(acc), len(acc), np.sum(np.asarray(ious) >= 0.25), np.sum(np.asarray(ious) >= 0.5),
                              len(ious)], dtype=torch.float64, device=self.device)
        if self.world > 1:
            dist.all_reduce(stats)                 # metrics over the whole validation set, not one shard
        s = stats.tolist()
        rec = {"epoch": epoch, "ref_acc": s[0] / max(s[1], 1), "iou_rate_0.25": s[2] / max(s[4], 1),
               "iou_rate_0.5": s[3] / max(s[4], 1)}
        self.log["val"].append(rec)
        self._scalars("val", rec)
        self._say("[val] epoch %d ref_acc %.4f Acc@0.25 %.4f Acc@0.5 %.4f" % (epoch, rec["ref_acc"], rec["iou_rate_0.25"],
                                                                           rec["iou_rate_0.5"]))
        return rec

    def __call__(self, epochs):
        for epoch in range(self.start_epoch, epochs):
            self.train_epoch(epoch)
            self.save("model_last.pth")
            rec = self.validate(epoch)
            if rec is not None and rec["iou_rate_0.25"] > self.best["iou_rate_0.25"]:
                self.best = dict(rec)
                self.save("model.pth")
        self.finish(epochs)

    # ------------------------------------------------------------------------------------------
    def save(self, name):
        if self.rank == 0 and self.out_dir:
            torch.save(self.model.state_dict(), os.path.join(self.out_dir, name))

    def finish(self, epoch):
        if self.rank == 0 and self.out_dir:
            torch.save({"epoch": epoch, "model_state_dict": self.model.state_dict(),
                        "optimizer_state_dict": self.optimizer.state_dict()},
                       os.path.join(self.out_dir, "checkpoint.tar"))
            with open(os.path.join(self.out_dir, "best.txt"), "w") as f:
                for k, v in self.best.items():
                    f.write("%s: %s\n" % (k, v))


class SyntheticLoader:
    """Iterable of reference-format batches from instancerefer_amd.synthetic (no network for the real dataset):
    `lidar` is built the way lib/dataset.py does — per-scene sparse_quantize + collate — but on the GPU."""

    def __init__(self, batches, batch_size, seed=123, rank=0, world=1, **scene_kw):
        self.batches, self.batch_size, self.seed, self.rank, self.world, self.kw = batches, batch_size, seed, rank, world, scene_kw

    def __iter__(self):
        from . import synthetic as S
        from .sparse.utils import voxelize
        dev = torch.device("cuda", torch.cuda.current_device())
        for b in range(self.batches):
            dd = S.make_batch(self.batch_size, seed=self.seed + (b * self.world + self.rank) * self.batch_size, **dict(self.kw))
            pts = [torch.from_numpy(p) for p in dd["scene_points"]]
            allp = torch.cat(pts, 0).to(dev)
            batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
            dd["lidar"] = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [0.05] * 3, len(pts))
            yield dd
