#!/usr/bin/env python3
"""Golden vectors for the SD VAE decoder: the REFERENCE's vendored `AutoencoderKL.decode` (sd/diffusers/src/diffusers/models/
autoencoders/autoencoder_kl.py:287-320) on CPU, fp32, for a narrow decoder (block_out_channels (64, 64, 128, 128)) and for the SD-1.5
width (128, 256, 512, 512) on a small latent -- with the weights of the product's own seeded initialiser
(diffusion_tts_amd.init.vae_decoder_state_dict) loaded into the reference module, so that nothing but inputs and outputs is stored.
Run: PYTHONHASHSEED=0 python tests/golden/make_golden_vae.py   (needs /root/reference).  Writes tests/golden/vae_golden.npz."""
import importlib.util
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

warnings.simplefilter('ignore')
import transformers
import transformers.utils
transformers.utils.FLAX_WEIGHTS_NAME = 'flax_model.msgpack'
REF = os.environ.get('DTS_REFERENCE', '/root/reference')
spec = importlib.util.spec_from_file_location('diffusers', os.path.join(REF, 'sd/diffusers/src/diffusers/__init__.py'))
sys.modules['diffusers'] = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sys.modules['diffusers'])
from diffusers import AutoencoderKL                                    # noqa: E402

from diffusion_tts_amd import init as dinit                            # noqa: E402

CASES = {'narrow': dict(boc=(64, 64, 128, 128), latent=(2, 4, 8, 8), seed=3),
         'sd15_width': dict(boc=(128, 256, 512, 512), latent=(1, 4, 16, 16), seed=4)}


def main():
    torch.set_num_threads(8)
    out = {}
    for name, c in CASES.items():
        vae = AutoencoderKL(in_channels=3, out_channels=3, down_block_types=('DownEncoderBlock2D',) * 4,
                            up_block_types=('UpDecoderBlock2D',) * 4, block_out_channels=c['boc'], layers_per_block=2,
                            latent_channels=4, norm_num_groups=32).eval()
        sd = dinit.vae_decoder_state_dict(c['boc'], 2, 4, seed=c['seed'])
        missing, unexpected = vae.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.startswith(('encoder.', 'quant_conv.')) for k in missing), (missing[:4], unexpected)
        g = torch.Generator().manual_seed(c['seed'] + 100)
        z = torch.randn(c['latent'], generator=g)
        with torch.no_grad():
            img = vae.decode(z, return_dict=False)[0]
        out[f'{name}_z'] = z.numpy()
        out[f'{name}_image'] = img.numpy()
        out[f'{name}_checksum'] = np.array([float(sum(v.double().abs().sum() for v in sd.values())), float(len(sd))])
        print(name, img.shape, float(img.abs().max()), float(img.std()))
    np.savez_compressed(os.path.join(HERE, 'vae_golden.npz'), **out)


if __name__ == '__main__':
    main()
