#!/usr/bin/env python3
"""Generation-time check of diffusion_tts_amd/checkpoint.py against pickles written by the REFERENCE's own persistence
machinery (runs only where /root/reference is importable: PYTHONHASHSEED=0 python tests/golden/check_pkl_loader.py).

For each net (tiny ADM, tiny DDPM++, full-size ADM-64 and DDPM++-32) the reference module is pickled IN MEMORY exactly as
`edm/training/training_loop.py` does (`pickle.dump(dict(ema=net))`; such a pickle embeds the reference's source text, so it is
never written to disk or committed), read back through `load_edm_pickle`, and compared with `net.state_dict()` and the
constructor arguments.  The outcome (key counts, checksums, versions) goes to tests/golden/pkl_loader_check.json, which the
CPU suite reads next to a synthetic pickle of the same layout (tests/test_checkpoint.py)."""
import io
import json
import os
import pickle
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                      # noqa: E402  (import recipe + reference constructors)
import torch                                  # noqa: E402

from diffusion_tts_amd import init as dinit   # noqa: E402
from diffusion_tts_amd.checkpoint import load_edm_pickle  # noqa: E402
from diffusion_tts_amd.config import adm_imagenet64, ddpmpp_cifar10  # noqa: E402


def main():
    report = {'torch': torch.__version__, 'nets': {}}
    cases = dict(mg.TINY)
    cases['adm_imagenet64'] = adm_imagenet64()
    cases['ddpmpp_cifar10'] = ddpmpp_cifar10()
    for name, cfg in cases.items():
        net = mg.ref_edm(cfg, 0)
        blob = io.BytesIO()
        pickle.dump(dict(ema=net, note='reference-written, in memory only'), blob)
        got_cfg, sd = load_edm_pickle(blob.getvalue())
        ref_sd = net.state_dict()
        assert list(sd.keys()) == list(ref_sd.keys()), (name, [k for k in ref_sd if k not in sd][:5], [k for k in sd if k not in ref_sd][:5])
        for k in ref_sd:
            assert torch.equal(sd[k], ref_sd[k]), (name, k)
        want = dict(cfg.__dict__)
        have = dict(got_cfg.__dict__)
        assert have == want, (name, {k: (have[k], want[k]) for k in want if have[k] != want[k]})
        params = {k: v for k, v in sd.items() if k in dict(net.named_parameters())}
        report['nets'][name] = dict(keys=len(sd), parameters=len(params), pickle_bytes=len(blob.getvalue()),
                                    checksum_parameters=dinit.checksum(params))
        print(name, 'ok', report['nets'][name])
    with open(os.path.join(HERE, 'pkl_loader_check.json'), 'w') as f:
        json.dump(report, f, indent=1)


if __name__ == '__main__':
    main()
