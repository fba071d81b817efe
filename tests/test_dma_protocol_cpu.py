"""Re-derivation of every hand-counted `s_waitcnt vmcnt(N)` of the split-fp16 3x3 / 1x1 kernel.

`conv_mfma_f16x3_kernel` (csrc/conv_mfma_f16x3.hip) keeps LDS-DMA transfers, untracked asm loads and output stores in flight
across barriers and waits for exactly the ones a phase needs with counted `vmcnt(N)` immediates: on gfx9 one counter covers
loads, stores and LDS-DMA, and they retire in program order.  The immediates are formulas over the tile's staging geometry
(weight-ring slots RING, weight pieces per wave and step PPW, activation pieces per wave and chunk APW, res loads RL, res group
RG).  This test replays one wave's program order -- prologue, K steps, chunk hand-overs, the folded res_conv phase, epilogues,
tile changes of a persistent workgroup -- as a queue of outstanding operations for EVERY instantiated tile (geometry from the
library itself: mi_debug_conv16_geometry) and every schedule shape the planner can produce, and asserts at each use that
  * the weight step about to be multiplied has landed, the activation chunk about to be transformed has landed, the res
    operands about to be split have landed;
  * a ring slot is refilled only after the step that read it (with the step barrier in between);
  * every immediate fits the 6-bit field.
It checks the protocol's arithmetic, not the compiler's output (tests/test_isa_audit_cpu.py looks at the ISA)."""
import ctypes as C
import itertools
import os
import re

import pytest

from midd_amd import native

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "medical-image-denoising-using-diffusion_amd", "csrc")


def instantiated_tiles():
    src = open(os.path.join(CSRC, "conv_mfma_f16x3.hip")).read()
    body = src[src.index("#define MIDD_CONV16_TILES(X)"):src.index("struct Tile16")]
    tiles = [tuple(int(v) for v in m) for m in re.findall(r"X\((\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\)", body)]
    assert len(tiles) >= 15
    return tiles


def geometry(ks, stride, tile, cb=0):
    ring, ppw, apw, lds = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = native.lib().mi_debug_conv16_geometry(ks, stride, *tile, cb, C.byref(ring), C.byref(ppw), C.byref(apw), C.byref(lds))
    if rc != 0:
        return None
    return ring.value, ppw.value, apw.value, lds.value


class Wave:
    """One wave's outstanding vector-memory operations, oldest first; they retire in order."""

    def __init__(self):
        self.q = []
        self.max_imm = 0

    def issue(self, label, n):
        self.q += [label] * n

    def wait(self, n):
        assert 0 <= n <= 63, f"vmcnt({n}) does not fit the 6-bit field"
        self.max_imm = max(self.max_imm, n)
        if len(self.q) > n:
            self.q = self.q[len(self.q) - n:]

    def landed(self, label):
        return label not in self.q


def replay(ks, ring, ppw, apw, mt, nt, wm, cb, nblk, res_steps, tiles_per_wg, has_resid):
    """Program order of conv_mfma_f16x3_kernel for one wave; returns the largest immediate used."""
    taps = ks * ks
    hsteps = (taps + 1) // 2
    D = ring - 1
    RL, RG = 2 * mt, (2 if mt == 1 else 1)
    nchunks = (nblk + cb - 1) // cb
    w = Wave()
    issued = [0]          # weight steps requested so far (W(k) goes to ring slot k % ring)
    consumed = [0]        # weight steps multiplied so far

    def issue_w():
        k = issued[0]
        assert k - ring <= consumed[0] - 1, f"ring slot of W({k}) refilled before step {k - ring} was read"
        w.issue(("W", k), ppw)
        issued[0] += 1

    def mfma_step():
        assert w.landed(("W", consumed[0])), f"weights of step {consumed[0]} multiplied before they landed (queue {w.q[:6]}...)"
        consumed[0] += 1

    def k_step(with_a, issue_next_a, next_label):
        w.wait((D - 1) * ppw + (apw if with_a else 0))        # then the step barrier (WM != 1)
        issue_w()
        if issue_next_a:
            w.issue(next_label, apw)
        mfma_step()

    def res_phase():
        if res_steps == 0:
            return
        g = 0
        w.issue(("R", g), RG * RL)
        w.wait(0)
        assert w.landed(("R", g))
        for r in range(0, res_steps, RG):
            more = r + RG < res_steps
            if more:
                w.issue(("R", g + 1), RG * RL)
                for i in range(RG):
                    if r + i < res_steps:
                        w.wait((D - 1) * ppw + (RG * RL if i < D else 0))      # the loads are younger than W(step) only for i < D
                        issue_w()
                        mfma_step()
                w.wait(RG * ppw)
                assert w.landed(("R", g + 1)), f"res operands of group {g + 1} split before they landed"
                g += 1
            else:
                for i in range(RG):
                    if r + i < res_steps:
                        w.wait((D - 1) * ppw)
                        issue_w()
                        mfma_step()

    def epilogue():
        if has_resid:
            w.issue("resid", mt * nt)
            w.wait(0)
        w.issue("store", mt * nt)          # compiler-tracked output stores: they stay in the in-order queue

    # prologue
    w.issue(("A", 0, 0), apw)
    for _ in range(D):
        issue_w()
    w.wait(0)
    for tile in range(tiles_per_wg):
        has_next_tile = tile + 1 < tiles_per_wg
        for c in range(nchunks):
            more_in_tile = c + 1 < nchunks
            more = more_in_tile or has_next_tile
            nxt = ("A", tile, c + 1) if more_in_tile else ("A", tile + 1, 0)
            full = cb == 2 and 2 * c + 1 < nblk
            steps = taps if full else hsteps
            assert w.landed(("A", tile, c)), f"chunk {c} of tile {tile} transformed before it landed"
            for j in range(steps):
                if more:
                    k_step(with_a=1 <= j <= D, issue_next_a=(j == 0), next_label=nxt)
                else:
                    k_step(False, False, None)
            if more:
                if not more_in_tile:
                    res_phase()
                after = steps - 1 + (0 if more_in_tile else res_steps)
                w.wait(D * ppw if after >= D else ppw if after == 1 else 0)
                assert w.landed(nxt), f"{nxt} transformed before it landed (after={after}, D={D})"
                if not more_in_tile:
                    epilogue()
            else:
                res_phase()
    w.wait(0)
    epilogue()
    return w.max_imm


def test_every_vmcnt_immediate_of_every_instantiated_tile():
    tiles = instantiated_tiles()
    checked, worst = 0, 0
    for tile in tiles:
        tw, mt, nt, wm, wn = tile
        variants = [(3, 1, 0), (3, 2, 0), (1, 1, 0)]
        if tile in ((16, 2, 3, 4, 1), (16, 1, 3, 4, 1)):
            variants.append((3, 1, 2))                                       # wide chunks
        for ks, stride, cbt in variants:
            geo = geometry(ks, stride, tile, cbt)
            if geo is None:
                continue                                                     # not instantiated (picker never reaches it)
            ring, ppw, apw, _ = geo
            cb = cbt if cbt else (2 if ks == 1 else 1)
            res_options = (0, 1, 2, 3, 4, 6, 9, 12) if (ks == 3 and stride == 1) else (0,)
            for nblk, res_steps, tiles_per_wg, has_resid in itertools.product((1, 2, 3, 6, 9, 24), res_options, (1, 2, 3), (False, True)):
                worst = max(worst, replay(ks, ring, ppw, apw, mt, nt, wm, cb, nblk, res_steps, tiles_per_wg, has_resid))
                checked += 1
    assert checked > 5000 and worst <= 63
    print(f"{checked} schedules replayed over {len(tiles)} tiles; largest vmcnt immediate {worst}")


def test_the_model_catches_an_overcounted_wait():
    """The checker itself.  A wait that allows MORE outstanding operations than are really younger than the data it needs returns
    early: here every step of a chunk counts the activation request (APW pieces) as younger than its weights, although from
    step D + 1 on that request is OLDER than the step's weight slot -- the step then multiplies weights that have not landed."""
    ring, ppw, apw, _ = geometry(3, 1, (16, 2, 3, 4, 1))
    D = ring - 1
    w = Wave()
    w.issue(("A", 0), apw)
    for k in range(D):
        w.issue(("W", k), ppw)
    w.wait(0)
    with pytest.raises(AssertionError):
        for step in range(D + 2):
            w.wait((D - 1) * ppw + apw)                 # correct for steps 1..D only
            w.issue(("W", D + step), ppw)
            if step == 0:
                w.issue(("A", 1), apw)
            assert w.landed(("W", step)), "weights multiplied before they landed"
