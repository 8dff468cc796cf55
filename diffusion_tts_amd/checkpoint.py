"""Weight ingestion from NVIDIA EDM network pickles -- the `network_pkl` the reference hands to `pickle.load(f)['ema']`
(edm/main.py:69-70) -- WITHOUT executing the Python source embedded in them.

On-disk format (edm/torch_utils/persistence.py:123-131, 185-208): a plain pickle of `{'ema': net, ...}` in which every
`@persistent_class` object (EDMPrecond, DhariwalUNet / SongUNet, UNetBlock, Conv2d, Linear, GroupNorm, ...) is reduced
to a call `torch_utils.persistence._reconstruct_persistent_obj(meta)` with
    meta = dict(type='class', version=6, module_src=<source text of networks.py>, class_name=..., state=<module __dict__>)
The reference rebuilds the classes by `exec(meta.module_src)`.  This reader never does: the reconstruct call is mapped to
a stub that keeps `class_name` and `state`, the module tree is walked through the standard `torch.nn.Module` state
(`_parameters`, `_buffers`, `_modules`), and the result is the (EDMConfig, state_dict) pair `networks.EDMPrecond` takes,
with the reference's own state-dict key names.  Globals are resolved through an exact (module, name) allow-list of data constructors; everything else is refused, and the
nested storage bytes are read by torch's restricted (`weights_only`) loader.

Scope = the hot path's nets: EDMPrecond over DhariwalUNet (ADM) or SongUNet with the DDPM++
options; VP/VE/iDDPM preconditioners and NCSN++ options raise NotImplementedError.
"""
import collections
import io
import pickle
from typing import Any, Dict, Tuple

import torch

from .config import EDMConfig

PERSISTENCE_VERSION = 6                      # edm/torch_utils/persistence.py:27


class Persisted:
    """What is kept of a persistent object: its class name and the `__dict__` the reference would restore."""

    def __init__(self, meta):
        if meta.get('type') != 'class':
            raise pickle.UnpicklingError(f'persistent object of type {meta.get("type")!r}')
        if meta.get('version') != PERSISTENCE_VERSION:
            raise pickle.UnpicklingError(f'persistence version {meta.get("version")!r}, expected {PERSISTENCE_VERSION}')
        self.class_name = meta['class_name']
        self.state = dict(meta['state'])         # module_src is dropped here, unread


class EasyDict(dict):
    """dnnlib.util.EasyDict stand-in (edm/dnnlib/util.py:41-55): a dict with attribute access."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def _safe_storage_from_bytes(b):
    """Stand-in for `torch.storage._load_from_bytes`, which is `torch.load(..., weights_only=False)` -- a second, unrestricted
    unpickle of bytes taken from the file.  Tensor storages are plain data: torch's own restricted loader reads them."""
    return torch.load(io.BytesIO(b), weights_only=True)


def _allowed_globals():
    """EXACT (module, name) pairs an EDM network pickle needs (listed from reference-written pickles, tests/golden/check_pkl_loader.py):
    tensor / parameter rebuilders, storage classes, the two stock containers, OrderedDict, numpy scalars/arrays (the training
    loop's bookkeeping entries).  A prefix rule ("anything under torch.*") is NOT safe: `torch.utils.collect_env.run`,
    `torch.load`, `torch.hub.load`, `numpy.load` ... all execute or unpickle attacker-chosen input."""
    import collections as _c
    import numpy as _np
    import torch._utils as _tu
    ok = {
        ('collections', 'OrderedDict'): _c.OrderedDict,
        ('torch._utils', '_rebuild_tensor_v2'): _tu._rebuild_tensor_v2,
        ('torch._utils', '_rebuild_parameter'): _tu._rebuild_parameter,
        ('torch.storage', '_load_from_bytes'): _safe_storage_from_bytes,
        ('torch', 'Size'): torch.Size,
        ('torch', 'device'): torch.device,
        ('torch.nn.modules.container', 'ModuleList'): torch.nn.ModuleList,
        ('torch.nn.modules.container', 'ModuleDict'): torch.nn.ModuleDict,
        ('numpy', 'ndarray'): _np.ndarray,
        ('numpy', 'dtype'): _np.dtype,
        ('_codecs', 'encode'): __import__('_codecs').encode,
    }
    for name in ('FloatStorage', 'HalfStorage', 'BFloat16Storage', 'DoubleStorage', 'LongStorage', 'IntStorage', 'ShortStorage',
                 'CharStorage', 'ByteStorage', 'BoolStorage'):
        ok[('torch', name)] = getattr(torch, name)
    for name in ('float32', 'float16', 'bfloat16', 'float64', 'int64', 'int32', 'int16', 'int8', 'uint8', 'bool'):
        ok[('torch', name)] = getattr(torch, name)
    try:                                               # numpy >= 2 pickles through numpy._core, older files name numpy.core
        from numpy._core import multiarray as _ma
    except Exception:                                  # pragma: no cover
        from numpy.core import multiarray as _ma
    for mod in ('numpy.core.multiarray', 'numpy._core.multiarray'):
        ok[(mod, '_reconstruct')] = _ma._reconstruct
        ok[(mod, 'scalar')] = _ma.scalar
    return ok


_SAFE_BUILTINS = {'set', 'frozenset', 'dict', 'list', 'tuple', 'slice', 'complex', 'bytearray', 'int', 'float', 'bool', 'str',
                  'bytes', 'range'}                   # constructors of plain data only (no getattr / eval / object)
_ALLOWED = None


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        global _ALLOWED
        if (module, name) == ('torch_utils.persistence', '_reconstruct_persistent_obj'):
            return Persisted
        if (module, name) == ('dnnlib.util', 'EasyDict'):
            return EasyDict
        if module == 'builtins' and name in _SAFE_BUILTINS:
            return getattr(__import__('builtins'), name)
        if _ALLOWED is None:
            _ALLOWED = _allowed_globals()
        try:
            return _ALLOWED[(module, name)]
        except KeyError:
            raise pickle.UnpicklingError(f'global {module}.{name} is not allowed in an EDM network pickle') from None


def _module_state(obj) -> Dict[str, Any]:
    if isinstance(obj, Persisted):
        return obj.state
    if isinstance(obj, torch.nn.Module):           # stock containers (ModuleDict) unpickle as themselves
        return obj.__dict__
    raise TypeError(f'not a module: {type(obj)}')


def _flatten(obj, prefix, out):
    """state_dict() of the pickled module tree (torch/nn/modules/module.py: parameters, persistent buffers, children)."""
    st = _module_state(obj)
    for name, p in (st.get('_parameters') or {}).items():
        if p is not None:
            out[prefix + name] = p.detach()
    skip = st.get('_non_persistent_buffers_set') or set()
    for name, b in (st.get('_buffers') or {}).items():
        if b is not None and name not in skip:
            out[prefix + name] = b.detach()
    for name, child in (st.get('_modules') or {}).items():
        if child is not None:
            _flatten(child, prefix + name + '.', out)


_ADM_DEFAULTS = dict(model_channels=192, channel_mult=[1, 2, 3, 4], channel_mult_emb=4, num_blocks=3, attn_resolutions=[32, 16, 8],
                     augment_dim=0)                                       # DhariwalUNet.__init__, networks.py:373-388
_SONG_DEFAULTS = dict(model_channels=128, channel_mult=[1, 2, 2, 2], channel_mult_emb=4, num_blocks=4, attn_resolutions=[16],
                      augment_dim=0, embedding_type='positional', channel_mult_noise=1, encoder_type='standard',
                      decoder_type='standard', resample_filter=[1, 1])    # SongUNet.__init__, networks.py:230-249
_DDPMPP_ONLY = dict(embedding_type='positional', channel_mult_noise=1, encoder_type='standard', decoder_type='standard',
                    resample_filter=[1, 1])


def _config(net: Persisted) -> EDMConfig:
    if net.class_name != 'EDMPrecond':
        raise NotImplementedError(f'preconditioner {net.class_name}: only EDMPrecond is on the search path (SURVEY.md section 8 a3)')
    st = net.state
    model = (st.get('_modules') or {}).get('model')
    if not isinstance(model, Persisted) or model.class_name not in ('DhariwalUNet', 'SongUNet'):
        raise NotImplementedError(f'denoiser {getattr(model, "class_name", type(model))}: DhariwalUNet or SongUNet expected')
    adm = model.class_name == 'DhariwalUNet'
    kw = dict(_ADM_DEFAULTS if adm else _SONG_DEFAULTS)
    recorded = model.state.get('_init_kwargs')
    if recorded is None:
        raise ValueError('the pickled denoiser did not record its constructor arguments (persistence.py:106-108)')
    kw.update({k: v for k, v in dict(recorded).items() if k in kw})
    if not adm:
        for k, v in _DDPMPP_ONLY.items():
            if list(kw[k]) != list(v) if isinstance(v, list) else kw[k] != v:
                raise NotImplementedError(f'SongUNet option {k}={kw[k]!r}: only the DDPM++ configuration is supported')
    return EDMConfig(arch='adm' if adm else 'ddpmpp', img_resolution=int(st['img_resolution']), img_channels=int(st['img_channels']),
                     label_dim=int(st['label_dim']), model_channels=int(kw['model_channels']), channel_mult=list(kw['channel_mult']),
                     channel_mult_emb=int(kw['channel_mult_emb']), num_blocks=int(kw['num_blocks']),
                     attn_resolutions=list(kw['attn_resolutions']), augment_dim=int(kw['augment_dim']),
                     sigma_data=float(st['sigma_data']), sigma_min=float(st['sigma_min']), sigma_max=float(st['sigma_max']))


def load_edm_pickle(source, key: str = 'ema') -> Tuple[EDMConfig, 'collections.OrderedDict[str, torch.Tensor]']:
    """`source`: path, bytes or binary file of an EDM network pickle.  Returns (EDMConfig, state_dict with reference keys)."""
    if isinstance(source, (bytes, bytearray)):
        f, close = io.BytesIO(source), False
    elif isinstance(source, str):
        f, close = open(source, 'rb'), True
    else:
        f, close = source, False
    try:
        data = _Unpickler(f).load()
    finally:
        if close:
            f.close()
    net = data[key] if isinstance(data, dict) else data
    if not isinstance(net, Persisted):
        raise ValueError(f'pickle entry {key!r} is {type(net)}, not a persistent network object')
    cfg = _config(net)
    sd = collections.OrderedDict()
    _flatten(net, '', sd)
    return cfg, sd
