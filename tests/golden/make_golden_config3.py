#!/usr/bin/env python3
"""BASELINE.json configs[2] END TO END through THE REFERENCE ITSELF: edm/main.py generate_image_grid, EPS_GREEDY, N = 64 candidates,
K = 4 iterations, 18 sigma steps (sigma_max 80 .. 0.002, S_churn 40), the full 192-wide ADM ImageNet-64 denoiser and the full
ImageNet-64 classifier scorer under the weight rule -- 8 995 denoiser rows, 72 decisions -- imported and run on the CPU of the build
container (about two hours on 8 cores).

Run:  PYTHONHASHSEED=0 python tests/golden/make_golden_config3.py [--seed S]      (needs /root/reference)

Writes tests/golden/config3_golden.npz + config3_manifest.json: the 72 reward vectors (64 each), the index the reference selected at
every decision, the final score, the final fp32 denoiser output (= the final state: the last step is an Euler step to sigma 0), the
uint8 image and the row count.  Inputs are the latents / label of fullsize_golden.npz (eg64_*); weights are re-created by
diffusion_tts_amd.init and pinned by the checksums in fullsize_manifest.json.  Nothing of the reference's text is stored.

The seed was chosen on the GPU beforehand (tools/seed_scan.py, profiles/r05_seed_scan.txt): among the seeds scanned it is the one whose
smallest non-zero top-2 reward gap is largest, so that a free-running comparison is not decided by a coin flip between two correct fp32
summation orders (reward noise ~1e-8).  The tests additionally walk the build along the recorded selections (forced_selections), which
keeps all 72 decisions comparable whatever the margins."""
import argparse
import json
import os
import sys
import time

assert os.environ.get('PYTHONHASHSEED') == '0', 'run with PYTHONHASHSEED=0 (edm/main.py:776 hashes strings)'

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                       # noqa: E402
import make_golden_fullsize as mf              # noqa: E402

import numpy as np                             # noqa: E402
import torch                                   # noqa: E402

from diffusion_tts_amd import init as dinit    # noqa: E402
from diffusion_tts_amd.config import ClassifierConfig, adm_imagenet64  # noqa: E402

PARAMS = dict(N=64, K=4, lambda_param=0.15, eps=0.4)
NUM_STEPS = 18


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seed', type=int, required=True)
    ap.add_argument('--steps', type=int, default=NUM_STEPS, help='(plumbing check only: anything but 18 is not config 3)')
    ap.add_argument('--K', type=int, default=PARAMS['K'])
    ap.add_argument('--prefix', default='config3', help='output name: <prefix>_golden.npz / <prefix>_manifest.json')
    ap.add_argument('--threads', type=int, default=max(1, min(8, os.cpu_count() or 1)))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    params, num_steps = dict(PARAMS, K=a.K), a.steps
    assert a.prefix != 'config3' or (num_steps == NUM_STEPS and params == PARAMS), 'config3_golden.npz is BASELINE configs[2] as it stands'
    t00 = time.time()
    full = np.load(os.path.join(HERE, 'fullsize_golden.npz'))
    with open(os.path.join(HERE, 'fullsize_manifest.json')) as f:
        fman = json.load(f)
    adm, adm_ck = mf.ref_full(adm_imagenet64(), mg.NET_SEED)
    ccfg = ClassifierConfig()
    cls = mg.ref_classifier(ccfg, mg.CLS_SEED)
    csd, _ = dinit.refill_degenerate(dinit.classifier_state_dict(ccfg, mg.CLS_SEED), mg.CLS_SEED)
    cls.load_state_dict(csd, strict=True)
    cls_ck = dinit.checksum(csd)
    for ck, ref in ((adm_ck['checksum'], fman['adm_imagenet64']['checksum']), (cls_ck, fman['cls_imagenet64']['checksum'])):
        # the very weights fullsize_golden.npz was made with (the f64 sums depend on the thread count in their last digits)
        assert ck['numel'] == ref['numel'] and all(abs(ck[k_] - ref[k_]) <= 1e-12 * abs(ref[k_]) for k_ in ('sum', 'abs_sum')), (ck, ref)
    inet = mg.ref_scorers.ImageNetScorer.__new__(mg.ref_scorers.ImageNetScorer)
    torch.nn.Module.__init__(inet)
    inet.model = cls
    lat = torch.from_numpy(full['eg64_latents'])
    lab = torch.eye(1000)[torch.from_numpy(full['eg64_label_idx']).long()]
    seen = []
    carried = []                                                     # per decision: what the reference's loop itself carried on (edm/main.py:842-857)
    prev = {}

    def ref_locals():
        f = sys._getframe(1)
        while f is not None and f.f_code.co_name != 'generate_image_grid':
            f = f.f_back
        assert f is not None
        return f.f_locals

    def note_carried(loc):
        """`new_pivot_noise` of the reference's frame is the noise its own `scores.argmax` picked at the PREVIOUS decision (:842-850); which
        candidate that was is read off by comparing it, bit for bit, with the rows of that decision's `all_noises` (:800)."""
        if 'cand' not in prev:
            return
        piv = loc['new_pivot_noise'].detach().double()
        hits = [n for n in range(prev['cand'].shape[0]) if torch.equal(prev['cand'][n], piv[0])]
        assert hits, 'the carried pivot is none of the previous candidates'
        v = piv.reshape(-1)
        carried.append(dict(index=hits[0], n_equal_rows=len(hits), sum=float(v.sum()), abs_sum=float(v.abs().sum()), head=v[:8].tolist(),
                            sha256=__import__('hashlib').sha256(piv.numpy().tobytes()).hexdigest()))

    def scorer(images, labels, timesteps):                           # progress: one line per decision
        loc = ref_locals()
        note_carried(loc)
        prev.clear()
        if images.shape[0] > 1:
            prev['cand'] = loc['all_noises'].detach().double().clone()
        s = inet(images, labels, timesteps)
        seen.append(s.detach().numpy().copy())
        if s.shape[0] > 1:
            srt = np.sort(seen[-1])[::-1]
            print(f'[{time.time() - t00:7.1f}s] decision {len(seen) - 1:2d}: argmax {int(seen[-1].argmax()):2d}, top-2 gap {float(srt[0] - srt[1]):.3e}', flush=True)
        return s
    # (edm/main.py's loop does not disable autograd; 64 full-size rows recorded on a CPU exceed this container's memory.  no_grad changes no value.)
    with torch.no_grad():
        lg, sl, png, err = mg.run_ref_search(adm, scorer, lat, lab, 'EPS_GREEDY', params, num_steps, seed=a.seed)
    assert err is None, err
    rew = [sc.numpy() for _, sc in sl.calls]
    nd = num_steps * params['K']
    assert [r.shape[0] for r in rew] == [64] * nd + [1], [r.shape for r in rew]
    rewards = np.stack(rew[:nd])
    srt = np.sort(rewards, axis=1)[:, ::-1]
    gaps = (srt[:, 0].astype(np.float64) - srt[:, 1].astype(np.float64)).tolist()
    # `selected` is what THE REFERENCE'S LOOP carried on (the candidate whose noise became its next pivot), not a re-derivation from the rewards
    assert len(carried) == nd, len(carried)
    selected = np.array([c['index'] for c in carried], dtype=np.int64)
    checks = dict(carried_pivot_matches_one_candidate=all(c['n_equal_rows'] == 1 for c in carried),
                  carried_equals_argmax_of_recorded_rewards=bool(np.array_equal(selected, rewards.argmax(axis=1))))
    print(checks, flush=True)
    out = dict(rewards=rewards, selected=selected, final_score=rew[nd], image=png,
               pivot_sum=np.array([c['sum'] for c in carried]), pivot_abs_sum=np.array([c['abs_sum'] for c in carried]),
               pivot_head=np.array([c['head'] for c in carried]),
               last_D=lg.calls[-1][2].numpy(), last_x=lg.calls[-1][0].numpy(), sigmas=np.array(sorted({float(c[1][0]) for c in lg.calls}, reverse=True)))
    man = dict(params=params, num_steps=num_steps, seed=a.seed, net_rows=int(sum(c[0].shape[0] for c in lg.calls)), scorer_calls=len(sl.calls),
               top2_gaps=gaps, selected=[int(v) for v in out['selected']], pivot_sha256=[c['sha256'] for c in carried], checks=checks,
               selected_source='the candidate whose noise the reference loop carried as new_pivot_noise (edm/main.py:846-857), matched bit for bit', exact_ties=int(sum(g == 0.0 for g in gaps)),
               min_nonzero_gap=min((g for g in gaps if g > 0), default=None), S=dict(S_churn=40, S_min=0.05, S_max=50, S_noise=1.003),
               latents='fullsize_golden.npz eg64_latents / eg64_label_idx', adm_imagenet64=adm_ck, cls_checksum=cls_ck,
               torch=torch.__version__, numpy=np.__version__, threads=a.threads, seconds=round(time.time() - t00, 1))
    np.savez_compressed(os.path.join(HERE, f'{a.prefix}_golden.npz'), **out)
    with open(os.path.join(HERE, f'{a.prefix}_manifest.json'), 'w') as f:
        json.dump(man, f, indent=1)
    print(f'[{time.time() - t00:7.1f}s] wrote {a.prefix}_golden.npz: {man["net_rows"]} rows, selected {man["selected"]}, min non-zero gap {man["min_nonzero_gap"]}')


if __name__ == '__main__':
    main()
