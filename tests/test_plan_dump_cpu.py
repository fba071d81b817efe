"""Host-only checks of the planner through `mi_debug_plan_dump` (no GPU, no finalize): which kernel instantiations, tiles, ring depths
and key splits the plans of the BASELINE.json shapes reach -- the inventory VERDICT r3 item 1a asked for, kept as a test so that a
planner change that moves a shipped configuration onto an instantiation no GPU test exercises is noticed on the CPU."""
import re
import sys
import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import plan_dump as pd

RANGE_KW = dict(model_channels=32, channel_mult=(1, 2), num_res_blocks=2, attention_resolutions=(1,), time_emb_dim=32)


def launches(kw, B, H, W, side, compute="f16x3"):
    text = pd.dump(pd.make_plan(kw, compute), B, H, W, side)
    rows = []
    for line in text.splitlines()[1:]:
        m = re.search(r"\| (midd::[^|]+?)\s*(?:\| grid (\d+)x(\d+) wgs/img (\d+) tiles (\d+)x(\d+) ring (\d+) ppw (\d+) apw (\d+) lds (\d+))?$", line)
        assert m, line
        rows.append(dict(line=line, kernel=m.group(1).strip(), ring=int(m.group(7)) if m.group(7) else None,
                         lds=int(m.group(10)) if m.group(10) else None, grid=(int(m.group(2)), int(m.group(3))) if m.group(2) else None))
    return text, rows


def test_default_network_launch_list_and_instantiations():
    """configs[1]: two sub-batch programs of 4 at 256x256 -- 73 launches each, and only these 3x3 instantiations."""
    text, rows = launches({}, 4, 256, 256, side=1)
    assert len(rows) == 73 and "ops=73" in text
    conv3 = sorted({r["kernel"] for r in rows if "conv_mfma_f16x3_kernel" in r["kernel"]})
    assert conv3 == sorted([
        "midd::conv_mfma_f16x3_kernel<3, 1, 16, 1, 3, 4, 1, false, 0>", "midd::conv_mfma_f16x3_kernel<3, 1, 16, 1, 3, 4, 1, true, 0>",
        "midd::conv_mfma_f16x3_kernel<3, 1, 16, 2, 3, 4, 1, false, 0>", "midd::conv_mfma_f16x3_kernel<3, 1, 16, 2, 3, 4, 1, true, 0>",
        "midd::conv_mfma_f16x3_kernel<3, 2, 16, 1, 3, 4, 1, false, 0>", "midd::conv_mfma_f16x3_kernel<3, 2, 16, 2, 3, 4, 1, false, 0>"]), conv3
    assert sum("attention_f16x3_kernel<96>" in r["kernel"] for r in rows) == 6
    # side-by-side programs keep every workgroup at three per CU (<= 53 KB) except the stride-2 tiles (one per CU)
    for r in rows:
        if r["lds"] is not None and "<3, 2," not in r["kernel"] and "conv_mfma" in r["kernel"]:
            assert r["lds"] <= 53 * 1024 + 512, r["line"]


def test_shipped_configurations_never_take_a_two_slot_ring():
    """The three 3x3 tiles with a two-slot weight ring and MT = 1 had their res-phase wait one step short until round 4
    (tests/test_dma_protocol_cpu.py); no plan of the default network -- any BASELINE batch / size, alone or side by side -- takes a
    two-slot ring at all (the GPU test `test_two_slot_ring_tiles_...` reaches them through a many-tiny-images shape)."""
    for B, S, side in [(1, 256, 0), (4, 256, 1), (8, 256, 0), (16, 256, 1), (32, 256, 0), (4, 512, 1), (8, 512, 0), (1, 512, 0), (2, 64, 1), (1, 64, 0)]:
        _, rows = launches({}, B, S, S, side)
        assert all(r["ring"] is None or r["ring"] == 0 or r["ring"] >= 4 for r in rows), (B, S, side, [r["line"] for r in rows if r["ring"] in (2, 3)])


def test_red_case_of_round_3_plan():
    """GPUTEST_r03's failing call: B = 4, 104x96 on the 32-channel topology runs two B = 2 programs -- every 3x3 on the 32-pixel two-wave
    tile (8,1,2,2,1) with ONE tile per workgroup, folded res_conv at 1 / 2 / 4 steps, key split 4 x 20 tiles, in_conv1_kernel (the kernel
    the defect was in: DESIGN.md section 2a)."""
    text, rows = launches(RANGE_KW, 2, 104, 96, side=1)
    assert "stat_rep=7" in text and "persist_wgs=640" in text
    k3 = {r["kernel"] for r in rows if "conv_mfma_f16x3_kernel<3, 1" in r["kernel"]}
    assert k3 == {"midd::conv_mfma_f16x3_kernel<3, 1, 8, 1, 2, 2, 1, false, 0>", "midd::conv_mfma_f16x3_kernel<3, 1, 8, 1, 2, 2, 1, true, 0>"}
    assert rows[0]["kernel"] == "midd::in_conv1_kernel"
    assert {int(v) for v in re.findall(r"res_steps(\d+)", text)} == {0, 1, 2, 4}
    assert "ksplit4 tps20" in text
    first = next(r for r in rows if "conv_mfma" in r["kernel"])
    assert first["grid"] == (624, 1) and "tiles 12x26" in first["line"] and "wgs/img 312" in first["line"]
