"""InstanceRefer — drop-in for the reference's models/instancerefer.py:14-70: same ctor
(`InstanceRefer(input_feature_dim, args)`), same plug-in mechanism (module names from the YAML config are
imported and must export LangModule / AttributeModule / RelationModule / SceneModule) and the same
`forward(data_dict) -> data_dict` contract."""
import importlib

import torch
import torch.nn as nn

_PKG = __name__.rsplit('.', 1)[0]
import os as _os
# Kernel maps, tile orders and the backward-only tables enqueued by prepare_finish() on the preparation stream (two native calls per
# pyramid) instead of lazily at the head of the encoders' streams / of their backward passes. Round 3 (ten k_kmap_s1 launches through
# per-level Python builders) measured +0.5 % resident and -5..15 % end to end, and it stayed off; round 6 (octree-descent maps, one
# call) measures bf16 4.287 -> 4.191, fp32 8.300 -> 8.189 ms per step: ON. IRX_PREP_TABLES=0 builds them lazily again.
_PREP_TABLES = _os.environ.get('IRX_PREP_TABLES', '1') == '1'
_PREP_PLANS = _os.environ.get('IRX_PREP_PLANS', '0') == '1'
# The language module (word projection, GRU, attention pooling, classifier: ~0.6 ms of dispatch per step, independent of both
# encoders) is issued by a helper THREAD on the same stream while this thread assembles and submits the encoders: ATen operators
# and C-ABI calls release the GIL, so the two overlap. The forward half of the bf16 step is host-paced:
# 5.83-5.88 -> 5.48-5.61 ms per step on one box, 5.57-5.64 -> 5.51-5.54 on a faster one; neutral in fp32 and at B = 32
# (GPU-paced). Training mode on a HIP device only; IRX_LANG_THREAD=0 issues it inline.
_LANG_THREAD = _os.environ.get('IRX_LANG_THREAD', '1') == '1'
_STREAMS_ENV = _os.environ.get('IRX_STREAMS')                     # dev A/B switch: '0' / '1' overrides the policy in _streams_ok
_g = _os.environ.get('IRX_BWD_GATE_ROWS', '2000').split(',')       # "rows" or "recorder rows,waiter rows"; 0: no gate between the encoders' backward passes
_BWD_GATE_ROWS = int(_g[0])
_BWD_GATE_WAIT_ROWS = int(_g[-1])
_BWD_GATE = _BWD_GATE_ROWS > 0
_STREAMS = _STREAMS_ENV != '0'                                   # three-stream training forward (_forward_streams); 0: round-4 layout
_PREBUILD_BWD = _os.environ.get('IRX_PREBUILD_BWD', '0') == '1'   # backward-only tables built behind the scene head: measured neutral (3004-3013 vs 2920-2998 scenes/s), off
_REL_THREAD = _os.environ.get('IRX_REL_THREAD', '0') == '1'      # dev: the relation head on that thread too (behind the language module)
# The scene encoder's backward can issue its weight gradients on the language stream beside its BatchNorm-backward / data-gradient
# chain (IRX_ENC_DC2 / IRX_ENC_WSTREAM, include/irx.h; bit-identical). Measured NEGATIVE at B = 16 bf16 (alternating runs on one box:
# 4.350 / 4.349 ms per step with it, 4.217 / 4.249 without — the two kernel families contend for the same CUs and the chain, which is
# the long pole, slows down); with two library-owned streams instead the step HALVED its speed (9.19 ms: a sixth / seventh stream
# shares a hardware queue with a busy one). OFF; IRX_WGRAD_LANG=1 enables it.
_WGRAD_LANG = int(_os.environ.get('IRX_WGRAD_LANG', '0'))          # 1: the scene encoder's; 2: both encoders'
# creation order of the two encoders' nodes = reverse order of their backward passes: 'sc' (default) issues the candidate encoder's
# backward first, then the scene encoder's; 'cs' the other way round (dev A/B)
_ATTACH_ORDER = ('_attr_encoded', '_scene_encoded') if _os.environ.get('IRX_ATTACH_ORDER', 'sc') == 'cs' else ('_scene_encoded', '_attr_encoded')
# Measured (round 6, B = 16 bf16, one box, alternating runs): attr_first + gate 3.829 ms per step, attr_first without the gate 4.242,
# scene_first without the gate 4.142, scene_first + gate 5.200 (the scene pass, issued first, polls for the candidate pass's mark)
_PREP_BEV = _os.environ.get('IRX_PREP_BEV', '1') == '1'           # dev A/B: the scene head's BEV tables built by the preparation stage
_GATE_TOKEN, _GATE_LOCK = [0], __import__('threading').Lock()
_HEAD_ORDER = _os.environ.get('IRX_HEAD_ORDER', 'attr_first')      # see _forward_streams (dev A/B)
_SEQ_BUMP = int(_os.environ.get('IRX_SEQ_BUMP', '256'))          # 0: leave the autograd sequence numbers alone (dev A/B)
MARK = None        # dev: bench.py's timeline mode installs a callable(name) here (phase marks inside forward)
_ATTR_EARLY = _os.environ.get('IRX_ATTR_EARLY')   # dev A/B switch: '0' / '1' overrides the policy in forward()


def _import(name):
    """YAML names are bare ('attribute_module'); resolve inside this package first, then globally."""
    try:
        return importlib.import_module(_PKG + '.' + name)
    except ModuleNotFoundError:
        return importlib.import_module(name)


class _LangWorker:
    """One daemon thread running (modules, data_dict) jobs posted by InstanceRefer._lang_async; see there."""

    def __init__(self):
        import queue
        import threading
        self.q_in, self.q_out, self.seq = queue.SimpleQueue(), queue.SimpleQueue(), 0
        self.thread = threading.Thread(target=_LangWorker._run, args=(self.q_in, self.q_out), name="irx-lang", daemon=True)
        self.thread.start()

    @staticmethod
    def _run(q_in, q_out):
        while True:
            job = q_in.get()
            if job is None:
                return
            seq, dev, stream, grad, mods, dd = job
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(stream), torch.set_grad_enabled(grad):     # (both are thread-local state)
                    for i, m in enumerate(mods):
                        if m is not None:
                            dd = m(dd)
                        if i == 1 and dd.get('_lang_event') is not None:
                            dd['_lang_event'].record(stream)             # language features (+ the heads' language MLPs) complete
                q_out.put((seq, dd))
            except BaseException as e:              # surfaced by the training thread at join time
                q_out.put((seq, e))
            del job, mods, dd                       # nothing of the model stays referenced between jobs

    def post(self, mods, dd, stream=None):
        self.seq += 1
        self.q_in.put((self.seq, torch.cuda.current_device(), stream if stream is not None else torch.cuda.current_stream(),
                       torch.is_grad_enabled(), mods, dd))
        return self.seq

    def take(self, seq):
        while True:
            got, out = self.q_out.get()
            if got == seq:
                return out                          # (older results: a join that was interrupted — dropped)

    def stop(self):
        self.q_in.put(None)


class InstanceRefer(nn.Module):
    def __init__(self, input_feature_dim=0, args=None):
        super().__init__()
        self.args = args
        self.lang = _import(args.language_module).LangModule(args.num_classes, True, args.use_bidir, 300, 128)
        if args.attribute_module:
            self.attribute = _import(args.attribute_module).AttributeModule(input_feature_dim, args)
        if args.relation_module:
            self.relation = _import(args.relation_module).RelationModule(input_feature_dim, args)
        if args.scene_module:
            self.scene = _import(args.scene_module).SceneModule(input_feature_dim, args)
        # the two sparse encoders are issued by library threads (sparse/encoder_fn.py "asynchronous issue"): lanes 0 / 1
        for lane, owner in enumerate((getattr(self, 'scene', None), getattr(self, 'attribute', None))):
            if owner is not None and hasattr(owner, 'net'):
                owner.net.__dict__['_irx_lane'] = lane

    def prepare(self, data_dict):
        """Phase 0 — everything that needs a host sync and depends only on the inputs: candidate selection +
        voxelisation + coordinate pyramid of the attribute path (needs the class list, i.e. GT classes), the relation
        node features, then Morton sort + pyramid of the scene tensor. forward() calls it when the caller has not; a
        training loop can call it for batch N+1 on a side stream while batch N's backward runs (bench.py does), after
        which forward() issues its work without ever draining the GPU queue. Returns data_dict (marked prepared)."""
        return self.prepare_finish(self.prepare_launch(data_dict))

    def prepare_launch(self, data_dict):
        """prepare() up to the waits for the two pyramids' level sizes: every kernel is enqueued, the sizes are on their
        way to pinned host memory. A loop that puts host work between prepare_launch() and prepare_finish() (bench.py:
        the whole issue of step N) gets those syncs for free."""
        if data_dict.get('_prepared') or data_dict.get('_prepare_launched'):
            return data_dict
        if self.args.attribute_module and self.args.use_gt_lang and hasattr(self.attribute, 'prepare_launch'):
            cls = data_dict['object_cat']
            cls_list = data_dict['_host']['object_cat'] if 'object_cat' in data_dict.get('_host', {}) else cls.tolist()
            cls_list = [int(v) for v in cls_list]
            data_dict = self.attribute.prepare_launch(data_dict, cls_list)
            if self.args.relation_module and hasattr(self.relation, 'prepare'):
                data_dict = self.relation.prepare(data_dict, cls_list)
        if self.args.scene_module and 'lidar' in data_dict and hasattr(data_dict['lidar'], 'pyramid') and hasattr(data_dict['lidar'], 'finish'):
            # a sparse.utils.VoxelizePending (scene_input with IRX_INPUT_VOXELIZE_LAUNCH=1): voxeliser + pyramid already enqueued,
            # sync-free; prepare_finish() collects the tensor
            data_dict['_scene_vox_pending'] = data_dict['lidar']
        elif self.args.scene_module and 'lidar' in data_dict:
            lidar = data_dict['lidar']
            if lidar._batch_size is None and 'point_min' in data_dict:
                lidar._batch_size = data_dict['point_min'].shape[0]
            lidar = lidar.canonical()
            data_dict['_scene_pending'] = lidar.level().build_pyramid_launch(4)
            data_dict['lidar'] = lidar
        data_dict['_prepare_launched'] = True
        return data_dict

    def prepare_finish(self, data_dict):
        if data_dict.get('_prepared'):
            return data_dict
        if not data_dict.get('_prepare_launched'):
            data_dict = self.prepare_launch(data_dict)
        if '_attr_pending' in data_dict:
            data_dict = self.attribute.prepare_finish(data_dict)
        pending = data_dict.pop('_scene_pending', None)
        vox = data_dict.pop('_scene_vox_pending', None)
        if vox is not None:
            data_dict['lidar'] = vox.finish()                # canonical, pyramid built (the one wait for voxel count + level sizes)
            pending = (None, None)
        c0 = getattr(getattr(self, 'attribute', None), 'input_feature_dim', 0)
        wide = 128 < c0 <= 136                               # the wide stem's weight gradient runs over pair lists (encoder_fn._uses_pairs)
        if pending is not None:
            if pending[0] is not None:
                data_dict['lidar'].level().build_pyramid_finish(pending)
            if self.training and _PREP_TABLES:
                # kernel maps, tile order and the backward-only tables (pair lists, transposed child maps) on the preparation stream
                data_dict['lidar'].level().build_tables(backward=torch.is_grad_enabled(), pairs_level0=wide)
            if self.training and _PREP_TABLES and _PREP_BEV and hasattr(getattr(self, 'scene', None), 'to_bev'):
                # the dense-BEV gather tables of the scene head (forward table + its transpose for the data gradient): coordinate-only
                lv = data_dict['lidar'].level()
                while lv._down is not None:
                    lv = lv._down.out_level
                bev = self.scene.to_bev[1]
                if lv.n > 0 and hasattr(bev, 'bev_shape'):
                    lv.bev_t(bev.bev_shape[0], bev.bev_shape[1], bev.n_kernels)
        prep = data_dict.get('_attr_prepared')
        if self.training and _PREP_TABLES and prep is not None and prep[0] is not None:
            prep[0].level().build_tables(backward=torch.is_grad_enabled(), pairs_level0=wide)
        if self.training and _PREP_PLANS and torch.is_grad_enabled():
            # dev (IRX_PREP_PLANS=1): the encoders' plans and level-only descriptors here instead of in the forward (this also
            # builds the kernel maps on the preparation stream, like IRX_PREP_TABLES)
            from .sparse import encoder_fn
            if self.args.scene_module and 'lidar' in data_dict and hasattr(self.scene, 'net'):
                encoder_fn.prebuild(self.scene.net, data_dict['lidar'].level())
            if prep is not None and prep[0] is not None and hasattr(self.attribute, 'net'):
                encoder_fn.prebuild(self.attribute.net, prep[0].level())
        data_dict['_prepared'] = True
        return data_dict

    @staticmethod
    def hand_over(data_dict, stream):
        """After prepare() ran on another stream: register its surviving tensors with `stream` (the compute stream)."""
        st = data_dict.get('lidar')
        if st is not None and hasattr(st, 'record_stream'):
            st.record_stream(stream)
        prep = data_dict.get('_attr_prepared')
        if prep is not None and prep[0] is not None:
            prep[0].record_stream(stream)
        if prep is not None and prep[1].get('_dev') is not None:
            for t in prep[1]['_dev'].values():
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(stream)
        rel = data_dict.get('_rel_prepared')
        if rel is not None and rel[0] is not None:
            for t in rel[0][2:]:
                t.record_stream(stream)
        lab = data_dict.get('_loss_prepared')
        if lab is not None and lab.get('buf') is not None:
            for t in (lab['buf'], lab['fbuf']):
                t.record_stream(stream)

    def _lang_async(self, data_dict, stream=None, rel=None):
        """The language module on a helper thread, same stream (see _LANG_THREAD above) -> join() returning its data_dict keys.
        Grad mode and the current stream are thread-local: both are handed over with the job. The worker holds no reference to
        this model (jobs carry the modules they run; a finalizer stops the thread when the model is collected), every job and
        result carries a sequence number (a result left behind by an interrupted join is discarded, never consumed by the next
        forward), and the worker's queues are not part of the module's pickled / deep-copied state (__getstate__)."""
        w = self.__dict__.get('_lang_worker')
        if w is None:
            w = self.__dict__['_lang_worker'] = _LangWorker()
            import weakref
            weakref.finalize(self, w.stop)
        if rel is None:
            rel = _REL_THREAD
        rel = bool(rel and self.args.relation_module and '_rel_prepared' in data_dict)
        orig = dict(data_dict)                          # what was there when the job was posted
        sub = dict(orig)                                # the worker adds / replaces keys in its own shallow copy
        from . import heads
        pre = heads.PreLang(self) if heads.pre_lang_ok(self, data_dict) else None     # the heads' language-side MLPs (heads.py)
        seq = w.post((self.lang, pre, self.relation if rel else None), sub, stream)    # (the worker records _lang_event behind slot 1)

        def join():
            out = w.take(seq)
            if isinstance(out, BaseException):
                raise out
            new = {k: v for k, v in out.items() if k not in orig or orig[k] is not v}   # what the worker produced
            if rel:
                data_dict.pop('_rel_prepared', None)    # consumed by the worker's copy
            return new, rel
        return join

    def __getstate__(self):
        """Module state without the per-process runtime objects (helper-thread queues, HIP streams): copy.deepcopy(model) and
        torch.save(model) work after a training forward; the copy creates its own lazily."""
        state = dict(self.__dict__)
        state.pop('_lang_worker', None)
        state.pop('_enc_streams', None)
        return state

    def _streams_ok(self, data_dict):
        """The three-stream forward needs the training configuration the bench / Solver run: every module present, candidates
        prepared from the GT classes (the encoders do not wait for the language module), a HIP device."""
        a = self.args
        if _STREAMS_ENV is None:
            # policy follows the compute dtype (measured on MI355X, B = 16, alternating runs): with bf16 operands the step is a
            # set of latency-bound chains and the three-stream layout is +0.5-2 %; in fp32 it is bound by the fp32-MFMA
            # convolutions, and the extra concurrency takes CUs from the scene encoder: 1816 -> 1664-1676 scenes/s (-8 %)
            from . import get_compute_dtype
            if get_compute_dtype() == 'fp32':
                return False
        return (_STREAMS and self.training and torch.is_grad_enabled() and a.attribute_module and a.relation_module
                and a.scene_module and data_dict['lang_feat'].is_cuda and 'lidar' in data_dict
                and getattr(a, 'overlap_streams', True) and hasattr(self.scene, 'head')
                and data_dict.get('_attr_prepared') is not None and data_dict['_attr_prepared'][0] is not None
                and data_dict.get('_rel_prepared') is not None and data_dict['_rel_prepared'][0] is not None)

    def _forward_streams(self, data_dict):
        """Training forward on three HIP streams (round 5). The step at B = 16 is a handful of long kernels (the two encoders)
        and ~400 short ones in dependent chains (heads, language module, loss): on one stream the chains add up — 0.7 ms of
        heads behind the encoders forward, 0.6 ms before the encoders' backward can start, 0.3 ms of GRU backward behind it
        (profiles/r05_timeline_*.txt). Here:
          side  : scene encoder -> scene head up to the scene vector / area classifier (SceneModule.head)
          lang  : language module -> relation head (neither needs an encoder), issued by the helper thread
          main  : candidate encoder -> attribute head -> scene scores (needs obj_feats + the scene vector) -> [loss]
        Autograd replays every node on its forward stream and orders gradients that cross streams, so the backward overlaps the
        same way: the scene encoder's backward starts behind the short scene head instead of behind every head, the GRU backward
        runs beside the encoders'. Same kernels, same arithmetic: bit-identical to the one-stream forward (tested)."""
        from .sparse.encoder_fn import lane_of, lane_wait
        main = torch.cuda.current_stream()
        dev = data_dict['lang_feat'].device
        side = self._encoder_stream(dev)
        # backward gate (irx_encoder_gate_next): the scene encoder's large levels start behind the candidate encoder's small ones
        if _BWD_GATE and _BWD_GATE_ROWS > 0:
            # a process-wide, monotonic token (per-model counters restarting at 1 let a second model's first waiter see the
            # first model's mark as its own and skip the gate)
            with _GATE_LOCK:
                _GATE_TOKEN[0] += 1
                tok = _GATE_TOKEN[0]
            self.scene.net._irx_bwd_gate = (2, _BWD_GATE_WAIT_ROWS, tok)
            self.attribute.net._irx_bwd_gate = (1, _BWD_GATE_ROWS, tok)
        else:
            self.scene.net._irx_bwd_gate = self.attribute.net._irx_bwd_gate = None
        if _SEQ_BUMP:
            self._bump_sequence()
        from . import heads
        heads.draw_seeds(data_dict, dev)                     # every dropout seed of the fused heads, in a fixed order, on this thread
        lstream = self._aux_stream(dev)
        # the scene encoder's backward (the step's long pole) can issue its weight gradients on the language stream, beside its own
        # BatchNorm-backward / data-gradient chain (sparse/encoder_fn.py WGRAD_STREAM; measured negative, off)
        self.scene.net.__dict__['_irx_wgrad_stream'] = lstream.cuda_stream if _WGRAD_LANG else None
        self.attribute.net.__dict__['_irx_wgrad_stream'] = lstream.cuda_stream if _WGRAD_LANG == 2 else None
        data_dict['_aux_stream'] = lstream                   # lent to the scene head's backward for its weight gradients (heads.py)
        side.wait_stream(main)                               # inputs and the optimizer's parameter update are complete
        lstream.wait_stream(main)
        self.hand_over(data_dict, lstream)                   # prepared tensors (relation node features, index lists) used on `lang`
        data_dict['_lang_event'] = torch.cuda.Event()
        if MARK: MARK("fwd: start")
        lang_join = self._lang_async(data_dict, stream=lstream, rel=True) if _LANG_THREAD else None
        try:
            lidar = data_dict['lidar']
            lidar.record_stream(side)
            with torch.cuda.stream(side):
                data_dict = self.scene.encode(data_dict)
            if MARK: MARK("fwd: scene encoder issued")
            data_dict = self.attribute.encode(data_dict, defer=True)
        except BaseException:
            if lang_join is not None:                        # never leave a result behind for the next forward to pick up
                try:
                    lang_join()
                except BaseException:
                    pass
            raise
        if MARK: MARK("fwd: encoders issued")
        if lang_join is not None:
            new, _ = lang_join()
            data_dict.update(new)
        else:
            with torch.cuda.stream(lstream):
                data_dict = self.lang(data_dict)
                if heads.pre_lang_ok(self, data_dict):
                    data_dict = heads.PreLang(self)(data_dict)
                data_dict['_lang_event'].record(lstream)
                data_dict = self.relation(data_dict)
        ev = data_dict.pop('_lang_event')
        if MARK: MARK("fwd: lang joined")
        for k in ('lang_attr_feats', 'lang_scene_feats', '_attr_lang_h', '_scene_lang_h'):   # produced on the language stream
            if isinstance(data_dict.get(k), torch.Tensor):
                data_dict[k].record_stream(main)
                data_dict[k].record_stream(side)
        # scene head on the encoder's stream (its launches were issued by a library thread: wait for that first)
        lane_wait(lane_of(self.scene.net))
        split = heads.attr_head_ok(self.attribute, self.scene, data_dict) is not None
        main.wait_event(ev)
        # Creation order of the nodes = reverse order of the backward (see _attach). _HEAD_ORDER 'scene_first': candidate encoder node,
        # attribute head, scene encoder node, scene head, scene scores, loss -> backward: loss, scene scores, scene head, SCENE ENCODER,
        # attribute head, candidate encoder. 'attr_first': scene encoder node, candidate encoder node, scene head, attribute head,
        # scene scores -> backward: ..., attribute head, scene head, candidate encoder, scene encoder.
        if split and _HEAD_ORDER == 'scene_first':
            self._attach(data_dict, '_attr_encoded')
            heads.attr_head(self.attribute, self.scene, data_dict)
            if MARK: MARK("fwd: attribute head issued")
            self._attach(data_dict, '_scene_encoded')
        else:
            for k in _ATTACH_ORDER:
                self._attach(data_dict, k)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            data_dict = self.scene.head(data_dict)
            # ... and, behind it, the tables only the encoders' backward passes need (pair lists, transposed maps): this stream
            # idles until the loss comes back, and they would otherwise head the backward chains
            if _PREBUILD_BWD:
                from .sparse.encoder_fn import prebuild_backward
                prebuild_backward(self.scene.net, data_dict['lidar'].level())
                prep = data_dict.get('_attr_prepared')
                if prep is not None and prep[0] is not None:
                    side.wait_stream(main)                   # the candidate levels' kernel maps were built on the main stream
                    prebuild_backward(self.attribute.net, prep[0].level(), use_stream=main)
        if MARK: MARK("fwd: scene head issued")
        if split and _HEAD_ORDER != 'scene_first':
            heads.attr_head(self.attribute, self.scene, data_dict)       # beside the scene head: it does not need the scene vector
            if MARK: MARK("fwd: attribute head issued")
        if not split:
            data_dict = self.attribute(data_dict)
            if MARK: MARK("fwd: attribute head issued")
        main.wait_stream(side)
        for k in ('_scene_feats', 'seg_scores', 'vis_atten'):
            data_dict[k].record_stream(main)
        if split:
            data_dict = heads.scene_scores(data_dict)        # cosine(vis_emb_fc1(obj_feats), scene vector): needs both heads
        else:
            data_dict = self.scene(data_dict)                # scene scores: needs obj_feats and the scene vector
        main.wait_stream(lstream)
        for k in ('relation_scores', 'lang_scores', 'lang_feat', 'atten_attr'):
            if isinstance(data_dict.get(k), torch.Tensor):
                data_dict[k].record_stream(main)
        data_dict.pop('_aux_stream', None)
        return data_dict

    @staticmethod
    def _attach(data_dict, key):
        """Create the autograd node of an encoder pass that was issued earlier (encoder_fn.Deferred) NOW. The engine runs ready nodes
        in reverse creation order: scene encoder node, candidate encoder node, scene head, attribute / scene-score head, loss are
        created in this order, so the backward issues loss -> attribute head -> scene head -> candidate encoder -> scene encoder and
        only then the relation head and the language module (whose nodes come from the helper thread: _bump_sequence). Measured
        (round 6, B = 16): with the candidate encoder AHEAD of the scene head the fp32 step loses 11 % (8.29 -> 9.25 ms: its kernels
        take CUs from the scene head's chain, which gates the scene encoder — the long pole), bf16 is indifferent (4.28 / 4.30)."""
        e = data_dict.get(key)
        if e is not None and hasattr(e, 'attach'):
            data_dict[key] = e.attach()

    @staticmethod
    def _bump_sequence():
        """Keep this thread's autograd sequence numbers ahead of the helper thread's (csrc/torch_nodes.cpp: bump_sequence)."""
        from . import _nodes
        mod = _nodes.load()
        if mod is not None and hasattr(mod, 'bump_sequence'):
            mod.bump_sequence(_SEQ_BUMP)

    def _aux_stream(self, device):
        cache = self.__dict__.setdefault('_enc_streams', {})
        st = cache.get((str(device), 'lang'))
        if st is None:
            st = cache[(str(device), 'lang')] = torch.cuda.Stream(device=device)
        return st

    def forward(self, data_dict):
        """reference models/instancerefer.py:37-70; the train-mode BatchNorm batch counters of the whole pass are bumped in one
        multi-tensor launch at its end (_counters.py)"""
        from . import _counters
        collect = self.training and data_dict['lang_feat'].is_cuda
        if collect:
            _counters.open_collector()
        try:
            return self._forward(data_dict)
        finally:
            if collect:
                _counters.close()

    def _forward(self, data_dict):
        data_dict = self.prepare(data_dict)
        if self._streams_ok(data_dict):
            return self._forward_streams(data_dict)
        for m in (getattr(self, 'scene', None), getattr(self, 'attribute', None)):
            if m is not None and hasattr(m, 'net'):
                m.net._irx_bwd_gate = None                   # (the gate belongs to the multi-stream layout)
                m.net.__dict__['_irx_wgrad_stream'] = None   # (... and so does the lent weight-gradient stream)
        side = None
        lang_join = None
        if _SEQ_BUMP and self.training and data_dict['lang_feat'].is_cuda:
            self._bump_sequence()
        if self.training and data_dict['lang_feat'].is_cuda:
            from . import heads as _h
            _h.draw_seeds(data_dict, data_dict['lang_feat'].device)
        if _LANG_THREAD and self.training and data_dict['lang_feat'].is_cuda:
            lang_join = self._lang_async(data_dict)
        try:
            data_dict, side = self._issue_encoders(data_dict)
        except BaseException:
            if lang_join is not None:                        # never leave a result behind for the next forward to pick up
                try:
                    lang_join()
                except BaseException:
                    pass
            raise
        rel_done = False
        if MARK: MARK("fwd: encoders issued")
        if lang_join is not None:
            new, rel_done = lang_join()
            data_dict.update(new)
        else:
            data_dict = self.lang(data_dict)
        if MARK: MARK("fwd: lang joined")
        from . import heads
        fused_tail = (self.args.attribute_module and self.args.scene_module and self.training and hasattr(self.scene, 'head')
                      and data_dict.get('_attr_prepared') is not None and data_dict['_attr_prepared'][0] is not None
                      and data_dict['lang_feat'].is_cuda and heads._mod() is not None)
        if self.args.attribute_module and not fused_tail:
            data_dict = self.attribute(data_dict)
        elif fused_tail:
            # the candidate encoder is issued HERE, where the per-operator attribute head issues it (beside the scene encoder, not
            # behind it: fp32 B = 16 measured 8.45 -> 10.16 ms per step when it waited for the scene head)
            data_dict = self.attribute.encode(data_dict, defer=True)
        if MARK: MARK("fwd: attribute head issued")
        if self.args.relation_module and not rel_done:
            data_dict = self.relation(data_dict)
        if MARK: MARK("fwd: relation head issued")
        if side is not None:
            from .sparse.encoder_fn import lane_of, lane_wait
            if MARK: MARK("fwd: before scene lane wait")
            lane_wait(lane_of(self.scene.net))               # every launch of the scene encoder is on `side` now
            if MARK: MARK("fwd: scene lane idle")
            main = torch.cuda.current_stream()
            main.wait_stream(side)
            data_dict['_scene_encoded'].record_stream(main)
        if fused_tail:
            # scene head, then attribute head + scene scores as one node (heads.py); whatever a precondition refuses runs per operator
            for k in _ATTACH_ORDER:
                self._attach(data_dict, k)
            data_dict = self.scene.head(data_dict)
            if heads.attr_head(self.attribute, self.scene, data_dict):
                data_dict = heads.scene_scores(data_dict)
            else:
                data_dict = self.attribute(data_dict)
                data_dict = self.scene(data_dict)
        elif self.args.scene_module:
            data_dict = self.scene(data_dict)
        return data_dict

    def _issue_encoders(self, data_dict):
        """The two sparse encoders, which depend on the prepared inputs only -> (data_dict, the scene encoder's stream or None)."""
        side = None
        if self.args.scene_module and 'lidar' in data_dict and hasattr(self.scene, 'encode'):
            # The whole-scene BEVEncoder depends on `lidar` only. It is issued FIRST and, on a HIP device, on its own
            # stream, so that it runs concurrently with the language / attribute / relation work of the main stream:
            # the deep levels of both sparse encoders are far too small to fill 256 CUs on their own, and autograd
            # replays each node's backward on its forward stream, so the two backward passes overlap as well.
            lidar = data_dict['lidar']
            if MARK: MARK("fwd: start")
            if lidar.F.is_cuda and getattr(self.args, 'overlap_streams', True):
                main = torch.cuda.current_stream()
                side = self._encoder_stream(lidar.F.device)
                side.wait_stream(main)                       # inputs and the optimizer's parameter update are complete
                lidar.record_stream(side)
                with torch.cuda.stream(side):
                    data_dict = self.scene.encode(data_dict)
            else:
                data_dict = self.scene.encode(data_dict)
        if MARK: MARK("fwd: scene encoder issued")
        if self.args.attribute_module and hasattr(self.attribute, 'encode') and self._attr_early():
            # candidates already chosen (prepare()): their encoder does not need the language features either, and its
            # launches are issued by a library thread while this one goes on with the language module (see _attr_early)
            data_dict = self.attribute.encode(data_dict, defer=True)
        return data_dict, side

    @staticmethod
    def _attr_early():
        """Issue the candidate encoder BEFORE the language module? Measured on MI355X (B = 16): with bf16 conv operands
        the step is host-bound and the early issue (a library thread launches the encoder while this thread issues
        the language module) is worth +15 % (1620 -> 1874 scenes/s, 1981 with both encoders asynchronous); in fp32 the
        step is GPU-bound and the early encoder takes CUs from the scene encoder on the critical path (-3 %)."""
        if _ATTR_EARLY is not None:
            return _ATTR_EARLY != '0'
        from . import get_compute_dtype
        return get_compute_dtype() != 'fp32'

    def _encoder_stream(self, device):
        """The scene encoder's stream: ONE per model and device, created on first use. Its priority follows the compute dtype in
        force at that moment (measured on MI355X, B = 16, alternating runs in one session): in fp32 the step is GPU-bound and a
        NORMAL-priority scene stream is +1.5 % (1652 vs 1627 scenes/s); with bf16 operands the scene encoder is the long pole
        and HIGH priority is +5 % (2056 vs 1960). IRX_ENC_PRIO overrides.
        Never a second stream after a dtype switch (round 5): the HIP runtime maps streams onto 4 hardware queues, this model
        already uses four (main, scene, language, the loop's preparation stream), and a fifth shares a queue with one of them —
        bench.py's second-dtype leg ran at HALF speed (fp32 992-1061 scenes/s in-process against 1816 in a process of its own)
        until the scene stream was reused."""
        cache = self.__dict__.setdefault('_enc_streams', {})     # not module attributes: never pickled with the state
        st = cache.get((str(device), 'scene'))
        if st is None:
            env = _os.environ.get('IRX_ENC_PRIO')
            if env is not None:
                prio = int(env)
            else:
                from . import get_compute_dtype
                prio = -1 if get_compute_dtype() != 'fp32' else 0
            st = cache[(str(device), 'scene')] = torch.cuda.Stream(device=device, priority=prio)
        return st
