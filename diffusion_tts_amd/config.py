"""Architecture descriptions for the two EDM denoisers and the scorer's classifier on the hot path.

The block enumeration follows the reference constructors so that parameter names and shapes are the
reference's own state-dict keys:
  DhariwalUNet.__init__  edm/training/networks.py:373-433   (ADM, ImageNet-64)
  SongUNet.__init__      edm/training/networks.py:230-318   (DDPM++ options only, CIFAR-10)
  EncoderUNetModel.__init__  edm/unet.py:708-870            (ImageNet-64 noisy classifier)
"""
from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass
class EDMConfig:
    arch: str = 'adm'                       # 'adm' | 'ddpmpp'
    img_resolution: int = 64
    img_channels: int = 3
    label_dim: int = 0
    model_channels: int = 192
    channel_mult: List[int] = field(default_factory=lambda: [1, 2, 3, 4])
    channel_mult_emb: int = 4
    num_blocks: int = 3
    attn_resolutions: List[int] = field(default_factory=lambda: [32, 16, 8])
    augment_dim: int = 0                    # DDPM++ CIFAR uses 9; parameter exists, input is None at inference
    sigma_data: float = 0.5
    sigma_min: float = 0.0
    sigma_max: float = float('inf')

    @property
    def emb_channels(self):
        return self.model_channels * self.channel_mult_emb


def adm_imagenet64(label_dim=1000) -> EDMConfig:
    """BASELINE config 3: edm/train.py:122-124 defaults for --arch=adm."""
    return EDMConfig('adm', 64, 3, label_dim, 192, [1, 2, 3, 4], 4, 3, [32, 16, 8], 0)


def ddpmpp_cifar10(label_dim=10) -> EDMConfig:
    """BASELINE configs 1-2: edm/train.py:116-118,146-147 defaults for --arch=ddpmpp."""
    return EDMConfig('ddpmpp', 32, 3, label_dim, 128, [2, 2, 2], 4, 4, [16], 9)


@dataclass
class Block:
    name: str
    kind: str                 # 'conv' | 'block'
    cin: int
    cout: int
    res_in: int
    res_out: int
    up: bool = False
    down: bool = False
    heads: int = 0


def edm_blocks(cfg: EDMConfig):
    """(encoder blocks, decoder blocks, final channel count), in construction order."""
    adm = cfg.arch == 'adm'
    mc = cfg.model_channels
    enc: List[Block] = []
    dec: List[Block] = []
    c = cfg.img_channels
    nlev = len(cfg.channel_mult)
    for lvl, mult in enumerate(cfg.channel_mult):
        r = cfg.img_resolution >> lvl
        if lvl == 0:
            c0 = mc * mult if adm else mc
            enc.append(Block(f'enc.{r}x{r}_conv', 'conv', c, c0, r, r))
            c = c0
        else:
            enc.append(Block(f'enc.{r}x{r}_down', 'block', c, c, 2 * r, r, down=True))
        for j in range(cfg.num_blocks):
            co = mc * mult
            att = r in cfg.attn_resolutions
            heads = (co // 64 if adm else 1) if att else 0
            enc.append(Block(f'enc.{r}x{r}_block{j}', 'block', c, co, r, r, heads=heads))
            c = co
    stack = [b.cout for b in enc]
    for lvl in range(nlev - 1, -1, -1):
        mult = cfg.channel_mult[lvl]
        r = cfg.img_resolution >> lvl
        if lvl == nlev - 1:
            dec.append(Block(f'dec.{r}x{r}_in0', 'block', c, c, r, r, heads=(c // 64 if adm else 1)))
            dec.append(Block(f'dec.{r}x{r}_in1', 'block', c, c, r, r))
        else:
            dec.append(Block(f'dec.{r}x{r}_up', 'block', c, c, r // 2, r, up=True))
        for j in range(cfg.num_blocks + 1):
            ci = c + stack.pop()
            co = mc * mult
            if adm:
                att = r in cfg.attn_resolutions
            else:
                att = (j == cfg.num_blocks) and (r in cfg.attn_resolutions)
            heads = (co // 64 if adm else 1) if att else 0
            dec.append(Block(f'dec.{r}x{r}_block{j}', 'block', ci, co, r, r, heads=heads))
            c = co
    return enc, dec, c


@dataclass
class ClassifierConfig:
    image_size: int = 64
    in_channels: int = 3
    model_channels: int = 128
    out_channels: int = 1000
    num_res_blocks: int = 4
    attention_ds: Tuple[int, ...] = (2, 4, 8)
    channel_mult: Tuple[int, ...] = (1, 2, 3, 4)
    num_head_channels: int = 64


@dataclass
class ClsLayer:
    kind: str                 # 'conv_in' | 'res' | 'attn'
    prefix: str
    cin: int
    cout: int
    res_in: int
    down: bool = False


def classifier_layers(cfg: ClassifierConfig):
    """Flat layer list (input_blocks then middle_block), final channels, final resolution."""
    mc = cfg.model_channels
    ch = int(cfg.channel_mult[0] * mc)
    res = cfg.image_size
    out: List[ClsLayer] = [ClsLayer('conv_in', 'input_blocks.0.0', cfg.in_channels, ch, res)]
    nblk, ds = 1, 1
    for lvl, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            co = int(mult * mc)
            out.append(ClsLayer('res', f'input_blocks.{nblk}.0', ch, co, res))
            ch = co
            if ds in cfg.attention_ds:
                out.append(ClsLayer('attn', f'input_blocks.{nblk}.1', ch, ch, res))
            nblk += 1
        if lvl != len(cfg.channel_mult) - 1:
            out.append(ClsLayer('res', f'input_blocks.{nblk}.0', ch, ch, res, down=True))
            nblk += 1
            ds *= 2
            res //= 2
    out.append(ClsLayer('res', 'middle_block.0', ch, ch, res))
    out.append(ClsLayer('attn', 'middle_block.1', ch, ch, res))
    out.append(ClsLayer('res', 'middle_block.2', ch, ch, res))
    return out, ch, res
