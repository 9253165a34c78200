"""Generate tests/golden/lang_variants.npz: the REFERENCE's models/lang_module.py LangModule (imported from /root/reference,
build container only) in the two constructor variants the YAML surface allows beside the default — `use_bidir: False`
(config/InstanceRefer.yaml; models/lang_module.py:8-49: a unidirectional 2-layer GRU, 128-d heads) and
`use_lang_classifier=False` (no lang_scores) — forward in eval mode plus the gradient of a fixed functional of the outputs
with respect to every parameter (train-mode dropout is off: p = 0). No stub is involved. Expected OUTPUTS only; inputs and
weights are regenerated from seeds.   Usage: python tests/golden/make_golden_lang_variants.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sys.path set-up + CPU shims)
from instancerefer_amd import synthetic as S  # noqa: E402
from helpers import WEIGHT_SEED  # noqa: E402

VARIANTS = {"unidir": (18, True, False, 300, 128), "nocls": (18, False, True, 300, 128), "unidir_nocls": (18, False, False, 300, 128)}
OUT_KEYS = ("lang_feat", "lang_cls_feats", "lang_attr_feats", "lang_rel_feats", "lang_scene_feats", "atten_attr", "atten_rel",
            "atten_scene")


def inputs():
    rng = np.random.default_rng(78)
    lens = np.array([30, 7, 126, 1, 64, 12])
    feat = np.zeros((6, 126, 300), np.float32)
    for i, L in enumerate(lens):
        feat[i, :L] = rng.standard_normal((L, 300)).astype(np.float32) * 0.4
    return feat, lens


STRIDE = 29      # fixtures stay small: elements 0, 29, 58, ... of the flattened tensor (coprime to every layer width)


def sample(a):
    return np.ascontiguousarray(a.reshape(-1)[::STRIDE])


def functional(dd, has_cls):
    """A fixed scalar of every output (so that every parameter gets a gradient)."""
    tot = 0.0
    for j, k in enumerate(("lang_cls_feats", "lang_attr_feats", "lang_rel_feats", "lang_scene_feats")):
        w = torch.linspace(-1.0, 1.0, dd[k].numel(), dtype=torch.float32, device=dd[k].device).view_as(dd[k])
        tot = tot + (dd[k] * w).sum() * (1.0 + 0.25 * j)
    if has_cls:
        tot = tot + (dd["lang_scores"] ** 2).sum()
    return tot


def main():
    MG.install_cpu_shims()
    from models.lang_module import LangModule
    feat, lens = inputs()
    out = {}
    for name, ctor in VARIANTS.items():
        lm = LangModule(*ctor)
        lm.load_state_dict(S.seeded_state_dict(lm, WEIGHT_SEED + 2))
        for m in lm.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        lm.train()
        dd = lm({"lang_feat": torch.from_numpy(feat), "lang_len": torch.from_numpy(lens)})
        assert ("lang_scores" in dd) == ctor[1]
        for k in OUT_KEYS + (("lang_scores",) if ctor[1] else ()):
            v = MG.t2n(dd[k])
            out["%s/%s" % (name, k)] = sample(v) if k == "lang_feat" else v
        functional(dd, ctor[1]).backward()
        for n, p in lm.named_parameters():                  # every parameter: a strided element sample + the norm
            g = MG.t2n(p.grad)
            out["%s/grad/%s" % (name, n)] = sample(g)
            out["%s/grad_norm/%s" % (name, n)] = np.float64(np.linalg.norm(g.astype(np.float64)))
    np.savez_compressed(os.path.join(HERE, "lang_variants.npz"), **out)
    print("lang_variants.npz:", len(out), "arrays,", sum(v.nbytes for v in out.values()), "bytes")


if __name__ == "__main__":
    main()
