"""Network description for the conditional noise-prediction UNet of the sampler path.

The reference builds its network imperatively in ``UNetDiffusion.__init__``
(/root/reference/Backend/DDIM/DDIMModel.py:169-217; cddpm variant:
/root/reference/Backend/cddpm/cddpmModels.py:176-232).  Here the same information is a
flat, declarative list of *module records* derived from the constructor arguments, so
that one description drives (a) the parameter container (state-dict names / shapes),
(b) the native planner in ``csrc/`` and (c) the CPU oracle used by the tests.

Nothing in this file touches torch or the GPU.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

GN_GROUPS = 8          # nn.GroupNorm(8, C) everywhere on the path (DDIMModel.py:116,121,139,214)
ATTN_HEADS = 2         # AttentionBlock(num_heads=2)            (DDIMModel.py:136)

VARIANT_DDIM = "ddim"      # Backend/DDIM/DDIMModel.py
VARIANT_CDDPM = "cddpm"    # Backend/cddpm/cddpmModels.py


@dataclass(frozen=True)
class UNetConfig:
    """Constructor arguments of the reference ``UNetDiffusion`` (DDIMModel.py:169-170)."""
    in_channels: int = 1
    model_channels: int = 48
    channel_mult: Tuple[int, ...] = (1, 2, 3, 4)
    num_res_blocks: int = 2
    attention_resolutions: Tuple[int, ...] = (3,)
    dropout: float = 0.0
    time_emb_dim: int = 192
    variant: str = VARIANT_DDIM

    def __post_init__(self):
        object.__setattr__(self, "channel_mult", tuple(int(c) for c in self.channel_mult))
        object.__setattr__(self, "attention_resolutions",
                           tuple(int(a) for a in self.attention_resolutions))
        if self.variant not in (VARIANT_DDIM, VARIANT_CDDPM):
            raise ValueError(f"unknown variant {self.variant!r}")
        if self.model_channels % 2 or self.model_channels < 4:
            raise ValueError("model_channels must be even and >= 4 (sinusoidal embedding halves it)")
        for m in self.channel_mult:
            if (self.model_channels * m) % GN_GROUPS:
                raise ValueError("every level width must be divisible by 8 (GroupNorm(8, C))")


@dataclass(frozen=True)
class Module:
    """One entry of ``downs`` / ``mid`` / ``ups`` (or in/out conv)."""
    kind: str        # 'rb' | 'attn' | 'down' | 'up'
    name: str        # state-dict prefix, e.g. 'downs.3', 'mid_block1', 'ups.6'
    in_c: int
    out_c: int


@dataclass
class Topology:
    cfg: UNetConfig
    downs: List[Module] = field(default_factory=list)
    mid: List[Module] = field(default_factory=list)
    ups: List[Module] = field(default_factory=list)
    final_c: int = 0

    @property
    def resblocks(self) -> List[Module]:
        """Residual blocks in execution order (this is the column order of the time table)."""
        return [m for m in self.downs + self.mid + self.ups if m.kind == "rb"]


def topology(cfg: UNetConfig) -> Topology:
    """Re-derive the module lists the reference constructor builds.

    DDIM variant: DDIMModel.py:182-211.  cddpm variant: cddpmModels.py:191-221 (it tracks
    the skip widths in ``down_channels`` and puts attention only after the first up block
    of an attention level).
    """
    topo = Topology(cfg)
    mc = cfg.model_channels
    nres = len(cfg.channel_mult)
    ch = mc
    down_channels: List[int] = []
    for i in range(nres):
        out_ch = mc * cfg.channel_mult[i]
        for _ in range(cfg.num_res_blocks):
            topo.downs.append(Module("rb", f"downs.{len(topo.downs)}", ch, out_ch))
            ch = out_ch
            down_channels.append(ch)
            if i in cfg.attention_resolutions:
                topo.downs.append(Module("attn", f"downs.{len(topo.downs)}", ch, ch))
                down_channels.append(ch)
        if i != nres - 1:
            topo.downs.append(Module("down", f"downs.{len(topo.downs)}", ch, ch))
            down_channels.append(ch)

    topo.mid = [Module("rb", "mid_block1", ch, ch),
                Module("attn", "mid_attn", ch, ch),
                Module("rb", "mid_block2", ch, ch)]

    for i in reversed(range(nres)):
        out_ch = mc * cfg.channel_mult[i]
        for j in range(cfg.num_res_blocks + 1):
            if cfg.variant == VARIANT_DDIM:
                in_ch = ch + ch                               # DDIMModel.py:205
            else:
                in_ch = ch + down_channels.pop()              # cddpmModels.py:216-217
            topo.ups.append(Module("rb", f"ups.{len(topo.ups)}", in_ch, out_ch))
            ch = out_ch
            if i in cfg.attention_resolutions and (cfg.variant == VARIANT_DDIM or j == 0):
                topo.ups.append(Module("attn", f"ups.{len(topo.ups)}", ch, ch))
        if i != 0:
            topo.ups.append(Module("up", f"ups.{len(topo.ups)}", ch, ch))
    topo.final_c = ch
    return topo


def param_shapes(cfg: UNetConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every state-dict tensor, in the reference's registration order.

    Names follow the reference's nn.Sequential indices: ``block1.0`` = GroupNorm,
    ``block1.2`` = Conv2d, ``block2.3`` = Conv2d (index 2 is Dropout), ``time_mlp.1`` =
    Linear (DDIMModel.py:111-126); ConvTranspose2d weights are [Cin, Cout, 4, 4].
    """
    topo = topology(cfg)
    mc, te = cfg.model_channels, cfg.time_emb_dim
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def conv(name, cin, cout, k):
        out.append((f"{name}.weight", (cout, cin, k, k)))
        out.append((f"{name}.bias", (cout,)))

    def gn(name, c):
        out.append((f"{name}.weight", (c,)))
        out.append((f"{name}.bias", (c,)))

    def linear(name, cin, cout):
        out.append((f"{name}.weight", (cout, cin)))
        out.append((f"{name}.bias", (cout,)))

    def emit(m: Module):
        if m.kind == "rb":
            linear(f"{m.name}.time_mlp.1", te, m.out_c)
            gn(f"{m.name}.block1.0", m.in_c)
            conv(f"{m.name}.block1.2", m.in_c, m.out_c, 3)
            gn(f"{m.name}.block2.0", m.out_c)
            conv(f"{m.name}.block2.3", m.out_c, m.out_c, 3)
            if m.in_c != m.out_c:
                conv(f"{m.name}.res_conv", m.in_c, m.out_c, 1)
        elif m.kind == "attn":
            gn(f"{m.name}.norm", m.in_c)
            conv(f"{m.name}.qkv", m.in_c, 3 * m.in_c, 1)
            conv(f"{m.name}.proj", m.in_c, m.in_c, 1)
        elif m.kind == "down":
            conv(m.name, m.in_c, m.out_c, 3)
        elif m.kind == "up":
            out.append((f"{m.name}.weight", (m.in_c, m.out_c, 4, 4)))
            out.append((f"{m.name}.bias", (m.out_c,)))

    linear("time_mlp.1", mc, te)
    linear("time_mlp.3", te, te)
    conv("in_conv", 2 * cfg.in_channels, mc, 3)
    for m in topo.downs:
        emit(m)
    for m in topo.mid:
        emit(m)
    for m in topo.ups:
        emit(m)
    gn("out_conv.0", topo.final_c)
    conv("out_conv.2", topo.final_c, cfg.in_channels, 3)
    return out


def timestep_list(noise_steps: int, inference_steps: int) -> List[int]:
    """Iteration list of the sampler: DDIMModel.py:272-274.

    ``reversed(range(0, noise_steps, max(1, noise_steps // inference_steps)))`` — note that
    inference_steps=8 with noise_steps=50 yields 9 iterations, and inference_steps larger
    than noise_steps still yields noise_steps iterations.
    """
    if inference_steps <= 0:
        raise ZeroDivisionError("integer division or modulo by zero")   # what the reference raises
    step = max(1, noise_steps // inference_steps)
    return list(reversed(range(0, noise_steps, step)))
