"""CPU: the VAE-decoder oracle (oracle/vae.py) against outputs of the reference's vendored AutoencoderKL.decode
(tests/golden/make_golden_vae.py), with the weights re-created by the product's seeded initialiser (pinned by checksum)."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from diffusion_tts_amd import init as dinit
from oracle import vae as ovae

CASES = {'narrow': ((64, 64, 128, 128), 3), 'sd15_width': ((128, 256, 512, 512), 4)}


@pytest.fixture(scope='module')
def vg():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'vae_golden.npz'))


@pytest.mark.parametrize('name', ['narrow', 'sd15_width'])
def test_vae_oracle_matches_reference_decoder(vg, name):
    boc, seed = CASES[name]
    sd = dinit.vae_decoder_state_dict(boc, 2, 4, seed=seed)
    ck = vg[f'{name}_checksum']
    assert len(sd) == int(ck[1]) and abs(float(sum(v.double().abs().sum() for v in sd.values())) - ck[0]) < 1e-9 * ck[0]
    torch.set_num_threads(8)
    with torch.no_grad():
        got = ovae.decode(sd, torch.from_numpy(vg[f'{name}_z']), block_out_channels=boc)
    want = torch.from_numpy(vg[f'{name}_image'])
    assert got.shape == want.shape and (got - want).abs().max().item() < 2e-4
