"""Weight ingestion from EDM network pickles: diffusion_tts_amd/checkpoint.py reads the layout
edm/torch_utils/persistence.py writes, never executing the source text such a file embeds."""
import json
import os
import pickle

import pytest
import torch

from helpers import tiny_edm, synthetic_edm_pickle
from diffusion_tts_amd.checkpoint import load_edm_pickle


@pytest.mark.parametrize('name', ['adm_tiny', 'ddpmpp_tiny'])
def test_reader_recovers_config_and_state_dict(manifest, name):
    cfg, sd = tiny_edm(manifest, name)
    blob = synthetic_edm_pickle(cfg, sd)
    got_cfg, got_sd = load_edm_pickle(blob)
    assert got_cfg == cfg
    assert list(got_sd.keys()) == list(sd.keys())
    for k in sd:
        assert torch.equal(got_sd[k], sd[k]), k


def test_reader_was_checked_against_reference_written_pickles():
    """tests/golden/check_pkl_loader.py (run where the reference is importable) pickles the reference's own modules in memory
    and compares; its report is committed."""
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'pkl_loader_check.json')) as f:
        rep = json.load(f)
    assert set(rep['nets']) >= {'adm_tiny', 'ddpmpp_tiny', 'adm_imagenet64', 'ddpmpp_cifar10'}
    assert rep['nets']['adm_imagenet64']['parameters'] == 541 and rep['nets']['adm_imagenet64']['checksum_parameters']['numel'] == 295899267


def test_reader_refuses_foreign_globals_and_other_preconditioners(manifest):
    evil = pickle.dumps(dict(ema=os.getcwd))                       # a global outside the allow-list
    with pytest.raises(pickle.UnpicklingError):
        load_edm_pickle(evil)
    cfg, sd = tiny_edm(manifest, 'adm_tiny')
    with pytest.raises(NotImplementedError):                       # VP/VE/iDDPM preconditioners are off the search path
        load_edm_pickle(synthetic_edm_pickle(cfg, sd, precond='VPPrecond'))


class _Call:
    """Pickles as a call of `fn(*args)` -- what a hostile network pickle would carry."""

    def __init__(self, fn, *args):
        self.fn, self.args = fn, args

    def __reduce__(self):
        return (self.fn, self.args)


@pytest.mark.parametrize('make', [
    lambda: _Call(__import__('torch.utils.collect_env', fromlist=['run']).run, 'echo PWNED'),     # shell command
    lambda: _Call(torch.load, '/nonexistent'),                                                     # nested full unpickle
    lambda: _Call(torch.hub.load, 'x', 'y'),
    lambda: _Call(__import__('numpy').load, '/nonexistent'),
    lambda: _Call(getattr, 'x', 'upper'),
    lambda: _Call(eval, '1+1'),
])
def test_reader_resolves_globals_by_exact_allow_list(make):
    """a prefix rule let `torch.utils.collect_env.run('echo PWNED')` execute.  Every global outside the exact
    allow-list of data constructors must raise before anything is called."""
    evil = pickle.dumps(dict(ema=make()))
    with pytest.raises(pickle.UnpicklingError):
        load_edm_pickle(evil)


def test_nested_storage_bytes_go_through_the_restricted_loader():
    """`torch.storage._load_from_bytes` is a plain `torch.load(weights_only=False)`; the reader substitutes torch's restricted
    loader, so a storage blob that itself carries a foreign global is refused too."""
    import io
    from diffusion_tts_amd import checkpoint as ck
    good = io.BytesIO()
    blob = pickle.dumps(torch.arange(4.))              # a plain-pickled tensor carries its storage as such bytes
    got = ck._Unpickler(io.BytesIO(pickle.dumps(dict(t=torch.arange(4.))))).load()['t']
    assert torch.equal(got, torch.arange(4.)) and blob
    with pytest.raises(Exception):
        ck._safe_storage_from_bytes(pickle.dumps(_Call(os.getcwd)))
