"""CPU: the oracle (oracle/) reproduces the golden vectors captured from the reference itself
(tests/golden/make_golden.py).  This is the pin that lets the GPU parity tests trust the oracle."""
import numpy as np
import pytest
import torch

from helpers import tiny_edm, tiny_cls, oracle_net, oracle_cls_cfg, T
from diffusion_tts_amd.hashing import seed0_scale
from oracle import sampler as osamp
from oracle import scorers as oscore
from oracle.classifier import encoder_unet, count_flops as cls_flops
from oracle.edm_nets import NetCfg, count_flops as net_flops

torch.set_num_threads(4)


@pytest.mark.parametrize('name', ['adm_tiny', 'ddpmpp_tiny'])
def test_denoiser_forward_matches_reference(golden, manifest, name):
    cfg, sd = tiny_edm(manifest, name)
    net = oracle_net(cfg, sd)
    labels = T(golden['fwd_labels'])
    for tag in ('hi', 'lo', 'rows'):
        x, s, D = (T(golden[f'fwd_{name}_{tag}_{k}']) for k in ('x', 'sigma', 'D'))
        got = net(x, s, labels)
        assert got.dtype == torch.float32
        err = (got - D).abs().max().item()
        assert err < 2e-5 * max(1.0, D.abs().max().item()), (tag, err)


def test_hash_scale_table(golden):
    tab = golden['hash_scale_table']
    mine = np.array([[[seed0_scale(i, k, n) for n in range(64)] for k in range(4)] for i in range(18)])
    assert np.array_equal(tab, mine)


def test_scorers(golden, manifest):
    img = T(golden['score_images'])
    b = oscore.BrightnessOracle()(img, None, torch.zeros(5))
    assert np.allclose(b.numpy(), golden['score_brightness'], atol=1e-7)
    cfg, sd = tiny_cls(manifest)
    ocfg = oracle_cls_cfg(cfg)
    logits = encoder_unet({k: v for k, v in sd.items()}, ocfg, img.float() / 255.0, torch.zeros(5))
    assert np.allclose(logits.numpy(), golden['score_cls_logits'], atol=2e-5)
    sc = oscore.ImageNetOracle(ocfg, sd)(img, T(golden['score_labels']), torch.zeros(5))
    assert np.allclose(sc.numpy(), golden['score_imagenet'], atol=1e-6)
    jp = oscore.CompressibilityOracle()(img, None, None)
    assert np.array_equal(jp.numpy(), golden['score_jpeg'])       # same Pillow build in this image


def _run_case(golden, manifest, case):
    meta = manifest['cases'][case]
    cfg, sd = tiny_edm(manifest, meta['net'])
    net = oracle_net(cfg, sd)
    if meta['scorer'] == 'brightness':
        scorer = oscore.BrightnessOracle()
    else:
        ccfg, csd = tiny_cls(manifest)
        scorer = oscore.ImageNetOracle(oracle_cls_cfg(ccfg), csd)
    b = meta['batch']
    lat, lab = T(golden[f'search_latents{b}']), T(golden[f'search_lab{b}'])
    method = {'NAIVE': 'naive', 'REJECTION_SAMPLING': 'rejection', 'EPS_GREEDY': 'eps_greedy',
              'ZERO_ORDER': 'zero_order', 'MCTS': 'mcts', 'BEAM_SEARCH': 'beam'}[meta['method']]
    np.random.seed(0)
    res = osamp.search(net, lat, lab, method=method, params=dict(scorer=scorer, **meta['params']), seed=meta['seed'],
                       num_steps=meta['num_steps'], S_churn=40, S_min=0.05, S_max=50, S_noise=1.003,
                       scale_fn=seed0_scale)
    return meta, net, res


@pytest.mark.parametrize('case', ['naive_adm', 'naive_ddpmpp', 'rejection_adm', 'rejection_ddpmpp',
                                  'epsgreedy_adm_bright', 'epsgreedy_adm_imagenet', 'zeroorder_adm', 'mcts_adm'])
def test_search_trace_matches_reference(golden, manifest, case):
    meta, net, res = _run_case(golden, manifest, case)
    assert net.evals == meta['net_rows']
    # every scorer call the reference made, in order; the last one is the final-image score
    n_calls = meta['scorer_calls']
    mine = res['rewards'] + [res['final_scores']]
    assert len(mine) == n_calls
    for j in range(n_calls):
        ref = golden[f'{case}_score{j}']
        got = mine[j].reshape(-1).numpy()
        assert np.allclose(got, ref.reshape(-1), atol=2e-6), (case, j, got, ref)
    # selected indices: recomputed from the reference's own reward vectors
    if meta['method'] in ('EPS_GREEDY', 'ZERO_ORDER'):
        N = meta['params']['N']
        for j, sel in enumerate(res['selected']):
            ref_sel = golden[f'{case}_score{j}'].reshape(N, meta['batch']).argmax(axis=0)
            assert np.array_equal(sel.numpy(), ref_sel), (case, j)
    if meta['method'] == 'REJECTION_SAMPLING':
        ref_sel = golden[f'{case}_score0'].reshape(meta['batch'], meta['params']['N']).argmax(axis=1)
        assert np.array_equal(res['selected'][0].numpy(), ref_sel)
    # final image (PNG written by the reference, read back): exact, allowing a rare 1-LSB flip at a
    # truncation boundary
    ref_img = golden[f'{case}_image']                      # [R, B*R, 3] grid, gridh=1
    R = ref_img.shape[0]
    mine_img = res['image'].permute(2, 0, 3, 1).reshape(R, -1, 3).numpy()
    diff = np.abs(mine_img.astype(int) - ref_img.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (case, diff.max(), (diff > 0).mean())


def test_beam_is_dead_code_like_the_reference(golden, manifest):
    assert 'AttributeError' in manifest['cases']['beam_adm']['error']
    with pytest.raises(AttributeError):
        _run_case(golden, manifest, 'beam_adm')


def test_flop_counters_match_reference_measurement(manifest):
    adm = net_flops(NetCfg('adm', 64, 3, 1000, 192, [1, 2, 3, 4], 4, 3, [32, 16, 8]))
    song = net_flops(NetCfg('ddpmpp', 32, 3, 10, 128, [2, 2, 2], 4, 4, [16], 9))
    cls = cls_flops(oracle_cls_cfg(__import__('diffusion_tts_amd.config', fromlist=['x']).ClassifierConfig()))
    assert abs(adm['total'] - manifest['adm_imagenet64']['flops']) / manifest['adm_imagenet64']['flops'] < 2e-3
    assert abs(song['total'] - manifest['ddpmpp_cifar10']['flops']) / manifest['ddpmpp_cifar10']['flops'] < 2e-3
    assert abs(cls['total'] - manifest['cls_imagenet64']['flops']) / manifest['cls_imagenet64']['flops'] < 2e-3


# ---- the oracle pinned AT FULL SIZE by outputs of the reference itself (tests/golden/make_golden_fullsize.py) -------
def test_fullsize_denoisers_match_reference(golden_full, manifest_full):
    """The 192-wide, 3-blocks-per-level, heads 6/9/12 ADM ImageNet-64 (networks.py:372-461) and the 128-wide DDPM++ CIFAR-32
    (networks.py:229-363): the oracle's 2-row forward against the REFERENCE module's on the same weights and inputs."""
    from helpers import full_weights
    torch.set_num_threads(8)
    for tag, which, L in (('adm64', 'adm_imagenet64', 1000), ('ddpmpp32', 'ddpmpp_cifar10', 10)):
        cfg, sd = full_weights(manifest_full, which)
        x, s, D = (T(golden_full[f'fwd_{tag}_{k}']) for k in ('x', 'sigma', 'D'))
        lab = torch.eye(L)[T(golden_full[f'fwd_{tag}_label_idx']).long()]
        got = oracle_net(cfg, sd)(x, s, lab)
        err = (got - D).abs().max().item() / max(1.0, D.abs().max().item())
        print(f'oracle vs reference, {which}: max err / max|D| = {err:.2e}')
        assert got.dtype == torch.float32 and err < 2e-5, (which, err)


def test_fullsize_classifier_and_scorer_match_reference(golden_full, manifest_full):
    """The w = 128, d = 4 ImageNet-64 classifier (unet.py:701-912) and ImageNetScorer.__call__ (scorers.py:143-174) at full size."""
    from helpers import full_weights
    torch.set_num_threads(8)
    cfg, sd = full_weights(manifest_full, 'cls_imagenet64')
    img = T(golden_full['cls64_images'])
    ocfg = oracle_cls_cfg(cfg)
    logits = encoder_unet(sd, ocfg, img.float() / 255.0, torch.zeros(2))
    ref = golden_full['cls64_logits']
    assert np.abs(logits.numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    lab = torch.eye(1000)[T(golden_full['cls64_label_idx']).long()]
    sc = oscore.ImageNetOracle(ocfg, sd)(img, lab, torch.zeros(2))
    assert np.allclose(sc.numpy(), golden_full['cls64_rewards'], rtol=1e-4, atol=1e-9)


class _StopAfterFirstDecision(Exception):
    pass


def test_fullsize_eps_greedy_n64_first_decision_matches_reference(golden_full, manifest_full):
    """ONE N = 64 eps-greedy iteration with the full ADM-64 denoiser and the full classifier: the oracle's 64 rewards and its argmax
    against what the reference's own generate_image_grid computed (edm/main.py:749-842) from the same latents, seed and hash table.
    (The oracle is stopped after the first scorer call: 128 denoiser rows + 64 classifier images, ~1.5 min on 8 cores; the GPU suite
    runs the whole 2-step search against the same golden.)"""
    from helpers import full_weights
    torch.set_num_threads(8)
    m = manifest_full['eg64']
    cfg, sd = full_weights(manifest_full, 'adm_imagenet64')
    ccfg, csd = full_weights(manifest_full, 'cls_imagenet64')
    inner = oscore.ImageNetOracle(oracle_cls_cfg(ccfg), csd)
    seen = []

    def scorer(images, labels, ts):
        seen.append(inner(images, labels, ts))
        raise _StopAfterFirstDecision

    lat = T(golden_full['eg64_latents'])
    lab = torch.eye(1000)[T(golden_full['eg64_label_idx']).long()]
    with pytest.raises(_StopAfterFirstDecision):
        osamp.search(oracle_net(cfg, sd), lat, lab, method='eps_greedy', params=dict(scorer=scorer, **m['params']), seed=m['seed'],
                     num_steps=m['num_steps'], sigma_max=m['sigma_max'], scale_fn=seed0_scale, **m['S'])
    got, ref = seen[0].numpy(), golden_full['eg64_rewards0']
    err = np.abs(got - ref).max()
    print(f'oracle vs reference, config-3 N=64 decision 0: max reward err {err:.2e}, reference top-2 gap {m["top2_gaps"][0]:.2e}, argmax {int(got.argmax())}')
    assert err < 5e-8 and m['top2_gaps'][0] > 4 * err
    assert int(got.argmax()) == int(golden_full['eg64_selected'][0]) == m['selected'][0]


@pytest.mark.parametrize('which', ['seed71', 'seed0'])
def test_config3_reference_run_golden_is_consistent_and_the_oracle_scores_its_final_image(golden_full, manifest_full, request, which):
    """tests/golden/config3_golden.npz (the reference's own end-to-end run of BASELINE configs[2]; the GPU suite compares whole searches with
    it): 72 decisions x 64 rewards, `selected` = first argmax, the exact ties are where no churn noise is drawn (sigma > 50, < 0.05), the 8 995
    denoiser rows are counted below, the PNG is the quantised final state (edm/main.py:869), and the ORACLE's
    denoiser + scorer reproduce the reference's last denoiser call and final score from the stored inputs (one full-size row each)."""
    from helpers import full_weights
    # config3_golden.npz: the run at the seed that was scanned on the GPU for wide margins (71); config3_seed0_golden.npz: a second run at an unscanned seed
    m, g = (request.getfixturevalue('manifest_c3'), request.getfixturevalue('golden_c3')) if which == 'seed71' else \
        (request.getfixturevalue('manifest_c3_seed0'), request.getfixturevalue('golden_c3_seed0'))
    assert m['seed'] == (71 if which == 'seed71' else 0)
    if 'pivot_sum' in g.files:          # the noise the reference's own loop carried on after each decision (make_golden_config3.py reads it off the loop's state)
        assert g['pivot_sum'].shape == (72,) and g['pivot_head'].shape == (72, 8) and len(m['pivot_sha256']) == 72 and len(set(m['pivot_sha256'])) == 72
        assert m['checks'] == dict(carried_pivot_matches_one_candidate=True, carried_equals_argmax_of_recorded_rewards=True)
    rew, sel = g['rewards'], [int(v) for v in g['selected']]
    assert rew.shape == (72, 64) and rew.dtype == np.float32 and sel == m['selected'] == [int(r.argmax()) for r in rew]
    # rows (edm/main.py:749-860): per sigma step K candidate batches of N rows + the batch-1 step of the final pivot, two denoiser calls each
    # (Heun), one at the last step (Euler to sigma = 0)
    N, K = m['params']['N'], m['params']['K']
    assert m['net_rows'] == 17 * 2 * (K * N + 1) + (K * N + 1) == 8995 and m['scorer_calls'] == 73
    sig = g['sigmas']
    assert len(sig) >= 18 and abs(sig[0] - 80.0) < 1e-9 and abs(sig[-1] - 0.002) < 1e-9      # every sigma a denoiser call saw, descending
    ties = [d for d in range(72) if m['top2_gaps'][d] == 0.0]
    for d in ties:
        assert np.all(rew[d] == rew[d][0]) and sel[d] == 0                      # identical candidates, first-max rule
    assert m['exact_ties'] == len(ties) and all(d < 8 or d >= 60 for d in ties)   # steps 0-1 (sigma 80, 57.6) and 15-17 (sigma < 0.05)
    a_, b_ = m['adm_imagenet64']['checksum'], manifest_full['adm_imagenet64']['checksum']
    assert a_['numel'] == b_['numel'] and abs(a_['abs_sum'] - b_['abs_sum']) <= 1e-12 * b_['abs_sum']
    # the final state is the last call's output (x_next = x_hat + (0 - t) * (x_hat - D) / t), the PNG its quantisation
    D = g['last_D'][0].astype(np.float64)
    img = np.clip(D * 127.5 + 128, 0, 255).astype(np.uint8).transpose(1, 2, 0)
    assert np.array_equal(img, g['image'])
    torch.set_num_threads(8)
    cfg, sd = full_weights(manifest_full, 'adm_imagenet64')
    ccfg, csd = full_weights(manifest_full, 'cls_imagenet64')
    lab = torch.eye(1000)[T(golden_full['eg64_label_idx']).long()]
    got = oracle_net(cfg, sd)(T(g['last_x']), torch.tensor([float(sig[-1])], dtype=torch.float64), lab)
    err = float((got - T(g['last_D'])).abs().max())
    sc = oscore.ImageNetOracle(oracle_cls_cfg(ccfg), csd)(torch.from_numpy(img.transpose(2, 0, 1)[None].copy()), lab, torch.zeros(1))
    print(f'oracle on the reference run\'s last step: max |D - D_ref| = {err:.2e}; final score {float(sc[0]):.6e} vs {float(g["final_score"][0]):.6e}')
    assert err < 2e-5 and abs(float(sc[0]) - float(g['final_score'][0])) < 5e-8


def test_mcts_fullsize_reference_run_golden_and_the_oracle(manifest_full, golden_mcts_full):
    """tests/golden/mcts_fullsize_golden.npz (the reference's own MCTS search at full network size, edm/main.py:405-713: N = 4, S = 16, three
    sigma steps): internal consistency, and the ORACLE's MCTS on the same inputs -- every group's rewards, the child made root at every
    timestep, the denoiser row count, the final state."""
    from helpers import full_weights
    from diffusion_tts_amd.hashing import seed0_scale
    g, m = golden_mcts_full
    assert g['rewards'].shape == (3, 16) and [int(v) for v in g['selected']] == m['selected'] and m['net_rows'] == 52 and m['scorer_calls'] == 4
    assert m['chosen'][0]['n_equal_in_value'] == 4 and m['chosen'][1]['n_equal_in_value'] == 1      # sigma 80 > S_max: no churn, the four children of the root are equal in value
    img = np.clip(g['x_final'][0] * 127.5 + 128, 0, 255).astype(np.uint8).transpose(1, 2, 0)
    assert np.array_equal(img, g['image'])
    torch.set_num_threads(8)
    cfg, sd = full_weights(manifest_full, 'adm_imagenet64')
    ccfg, csd = full_weights(manifest_full, 'cls_imagenet64')
    lat = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(m['latent_seed']))
    assert np.array_equal(lat.numpy(), g['latents'])
    lab = torch.eye(1000)[torch.tensor([m['label']])]
    onet = oracle_net(cfg, sd)
    np.random.seed(m['numpy_seed'])
    o = osamp.search(onet, lat, lab, method='mcts', params=dict(scorer=oscore.ImageNetOracle(oracle_cls_cfg(ccfg), csd), **m['params']), scale_fn=seed0_scale,
                     seed=m['seed'], num_steps=m['num_steps'], **m['S'])
    errs = [float(np.abs(o['rewards'][j].reshape(-1).numpy().astype(np.float64) - g['rewards'][j].astype(np.float64)).max()) for j in range(3)]
    x_err = float((o['x'] - T(g['x_final'])).abs().max())
    print(f'oracle MCTS at full size vs the reference run: reward errs {errs}, children {[int(v) for v in o["selected"]]} (reference {m["selected"]}), rows {onet.evals}, max |x - x_ref| {x_err:.2e}')
    assert max(errs) < 5e-8 and [int(v) for v in o['selected']] == m['selected'] and onet.evals == m['net_rows'] and x_err < 1e-3


def test_configs01_reference_runs_golden_and_the_oracle(manifest_full):
    """tests/golden/configs01_golden.npz (the reference's own runs of BASELINE configs[0] and [1] at full DDPM++ CIFAR-32 size): internal consistency
    (images = the quantised final states, the kept rejection trajectory = the argmax of the 16 recorded rewards), and the ORACLE's naive sampler on
    configs[0] (35 full-size rows) reproduces the reference's final state."""
    import json
    import os
    from conftest import ROOT
    from helpers import full_weights
    gp = os.path.join(ROOT, 'tests', 'golden', 'configs01_golden.npz')
    if not os.path.exists(gp):
        pytest.skip('tests/golden/configs01_golden.npz not generated')
    g = np.load(gp)
    with open(os.path.join(ROOT, 'tests', 'golden', 'configs01_manifest.json')) as f:
        m = json.load(f)
    for tag in ('naive', 'rej'):
        img = np.clip(g[f'{tag}_x_final'][0] * 127.5 + 128, 0, 255).astype(np.uint8).transpose(1, 2, 0)
        assert np.array_equal(img, g[f'{tag}_image'])
    assert m['rejection']['kept'] == m['rejection']['argmax_of_rewards'] == int(g['rej_rewards'].argmax()) == int(g['rej_kept'][0]) and m['rejection']['rows_with_that_image'] == 1
    assert m['naive']['net_rows'] == 35 and m['rejection']['net_rows'] == 16 * 35
    torch.set_num_threads(8)
    cfg, sd = full_weights(manifest_full, 'ddpmpp_cifar10')
    lat = torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(m['naive']['latent_seed']))
    o = osamp.search(oracle_net(cfg, sd), lat, torch.eye(10)[torch.tensor([m['naive']['label']])], method='naive', params=dict(scorer=oscore.BrightnessOracle()),
                     seed=m['seed'], num_steps=m['num_steps'], **m['S'])
    err = float((o['x'] - T(g['naive_x_final'])).abs().max())
    print(f'oracle naive sampler at full DDPM++-32 size vs the reference run: max |x - x_ref| = {err:.2e}')
    assert err < 1e-4 and (o['image'].int() - T(g['naive_image']).permute(2, 0, 1)[None].int()).abs().max().item() <= 1
