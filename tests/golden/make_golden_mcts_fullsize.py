#!/usr/bin/env python3
"""BASELINE.json configs[4] (EDM ImageNet-64 MCTS, imagenet scorer) AT FULL NETWORK SIZE through THE REFERENCE ITSELF: edm/main.py
generate_image_grid, SamplingMethod.MCTS (edm/main.py:405-713), the full 192-wide ADM ImageNet-64 denoiser and the full ImageNet-64
classifier scorer under the weight rule; N = 4 children per node, S = 16 simulations per timestep (one group of 16, edm/main.py:518),
three sigma steps, torch seed 1, np.random.seed(0) -- the case tests/test_gpu_fullsize.py::test_config5_mcts_full_size_* runs.

Run:  PYTHONHASHSEED=0 python tests/golden/make_golden_mcts_fullsize.py      (needs /root/reference; a few minutes of batch-1 CPU forwards)

Writes tests/golden/mcts_fullsize_golden.npz + mcts_fullsize_manifest.json: per group of simulations the 16 rewards the reference's
scorer returned; per timestep the child the REFERENCE'S LOOP made its next root (read off its own state: the key of its new root looked up
among the children of the old root, edm/main.py:684-703); the denoiser row count; the final state and uint8 image.  Inputs:
latents = randn(1, 3, 64, 64) from torch.Generator().manual_seed(6), label 417; weights re-created by diffusion_tts_amd.init and pinned
by the checksums of fullsize_manifest.json.  Nothing of the reference's text is stored."""
import json
import os
import sys
import time

assert os.environ.get('PYTHONHASHSEED') == '0', 'run with PYTHONHASHSEED=0'

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                       # noqa: E402
import make_golden_fullsize as mf              # noqa: E402

import numpy as np                             # noqa: E402
import torch                                   # noqa: E402

from diffusion_tts_amd import init as dinit    # noqa: E402
from diffusion_tts_amd.config import ClassifierConfig, adm_imagenet64  # noqa: E402

PARAMS = dict(N=4, S=16)
NUM_STEPS, SEED, LATENT_SEED, LABEL = 3, 1, 6, 417


def main():
    torch.set_num_threads(int(os.environ.get('DTS_GOLDEN_THREADS', max(1, min(8, os.cpu_count() or 1)))))
    t00 = time.time()
    with open(os.path.join(HERE, 'fullsize_manifest.json')) as f:
        fman = json.load(f)
    adm, adm_ck = mf.ref_full(adm_imagenet64(), mg.NET_SEED)
    ccfg = ClassifierConfig()
    cls = mg.ref_classifier(ccfg, mg.CLS_SEED)
    csd, _ = dinit.refill_degenerate(dinit.classifier_state_dict(ccfg, mg.CLS_SEED), mg.CLS_SEED)
    cls.load_state_dict(csd, strict=True)
    cls_ck = dinit.checksum(csd)
    for ck, ref in ((adm_ck['checksum'], fman['adm_imagenet64']['checksum']), (cls_ck, fman['cls_imagenet64']['checksum'])):
        assert ck['numel'] == ref['numel'] and all(abs(ck[k_] - ref[k_]) <= 1e-12 * abs(ref[k_]) for k_ in ('sum', 'abs_sum')), (ck, ref)
    inet = mg.ref_scorers.ImageNetScorer.__new__(mg.ref_scorers.ImageNetScorer)
    torch.nn.Module.__init__(inet)
    inet.model = cls
    lat = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(LATENT_SEED))
    lab = torch.eye(1000)[torch.tensor([LABEL])]
    chosen, state = [], {}

    def ref_locals():
        f = sys._getframe(1)
        while f is not None and f.f_code.co_name != 'generate_image_grid':
            f = f.f_back
        assert f is not None
        return f.f_locals

    def note_root(loc):
        """The reference keeps its tree in dicts keyed by tensor address (edm/main.py:433-435); when the root key has changed since the last
        scorer call, the new root is a clone of the child its own loop picked (:699-703): find that child among the old root's children."""
        key = loc['all_root_keys'][0]
        if 'key' in state and key != state['key']:
            kids = loc['all_children'][0][state['key']]
            # by KEY (the new root key is the picked child's own key, :703): at a timestep without churn noise (sigma > S_max) the children
            # are equal in value and only their identity says which subtree the search keeps
            hits = [j for j, (_, ck) in enumerate(kids) if ck == key]
            assert len(hits) == 1, 'the new root key is not exactly one of the old root\'s children'
            same = sum(int(torch.equal(child, loc['all_roots'][0])) for child, _ in kids)
            assert torch.equal(kids[hits[0]][0], loc['all_roots'][0])
            chosen.append(dict(child=hits[0], n_equal_in_value=same, n_children=len(kids)))
        state['key'] = key

    def scorer(images, labels, timesteps):
        loc = ref_locals()
        note_root(loc)
        s = inet(images, labels, timesteps)
        print(f'[{time.time() - t00:7.1f}s] scorer call: {tuple(images.shape)} -> rewards {s.detach().numpy().round(6).tolist()[:4]}...', flush=True)
        if images.shape[0] == 1 and 'x_next' in loc and len(chosen) == NUM_STEPS:
            state['x_final'] = loc['x_next'].detach().double().clone()
        return s
    with torch.no_grad():
        lg, sl, png, err = mg.run_ref_search(adm, scorer, lat, lab, 'MCTS', PARAMS, NUM_STEPS, seed=SEED)      # (np.random.seed(0) inside)
    assert err is None, err
    groups = [sc.numpy() for _, sc in sl.calls[:-1]]
    assert all(g.shape == (16,) for g in groups) and len(groups) == NUM_STEPS, [g.shape for g in groups]
    assert len(chosen) == NUM_STEPS, chosen
    out = dict(rewards=np.stack(groups), selected=np.array([c['child'] for c in chosen], dtype=np.int64), final_score=sl.calls[-1][1].numpy(),
               image=png, x_final=state['x_final'].numpy(), latents=lat.numpy(), label_idx=np.array([LABEL]),
               sigmas=np.array(sorted({float(c[1][0]) for c in lg.calls}, reverse=True)))
    man = dict(params=PARAMS, num_steps=NUM_STEPS, seed=SEED, numpy_seed=0, latent_seed=LATENT_SEED, label=LABEL,
               net_rows=int(sum(c[0].shape[0] for c in lg.calls)), net_calls=len(lg.calls), scorer_calls=len(sl.calls),
               selected=[int(v) for v in out['selected']], chosen=chosen, S=dict(S_churn=40, S_min=0.05, S_max=50, S_noise=1.003),
               selected_source='the child whose state the reference loop cloned into its next root (edm/main.py:699-703), matched by its node key',
               adm_imagenet64=adm_ck, cls_checksum=cls_ck, torch=torch.__version__, numpy=np.__version__, threads=torch.get_num_threads(),
               seconds=round(time.time() - t00, 1))
    np.savez_compressed(os.path.join(HERE, 'mcts_fullsize_golden.npz'), **out)
    with open(os.path.join(HERE, 'mcts_fullsize_manifest.json'), 'w') as f:
        json.dump(man, f, indent=1)
    print(f'[{time.time() - t00:7.1f}s] wrote mcts_fullsize_golden.npz: {man["net_rows"]} rows in {man["net_calls"]} calls, chosen children {man["selected"]}')


if __name__ == '__main__':
    main()
