"""CPU-side checks: the library loads and exports the declared C ABI, the engine's host logic
(parameter table, flat layout, launch plan, FLOP accounting) agrees with the oracle's module tree,
and the product fails loudly where it has no GPU."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

import oracle
import unet_amd
from unet_amd import _lib
from unet_amd.nets import _Plan

ROOT = Path(__file__).resolve().parent.parent
CFG = dict(spatial_dims=2, in_channels=1, out_channels=1, channels=(16, 32, 64, 128, 256), strides=(2, 2, 2, 2),
           num_res_units=2)


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "unet_mi355.h").read_text()
    declared = set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    lib = ctypes.CDLL(str(_lib.lib_path()))
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert _lib.load().mi_version() >= 100


@pytest.mark.parametrize("dims,channels", [(2, (16, 32, 64, 128, 256)), (2, (32, 64, 128, 256, 512)),
                                           (3, (16, 32, 64, 128, 256)), (2, (16, 32)), (2, (16, 32, 64))])
def test_param_table_matches_oracle_state_dict(dims, channels):
    strides = (2,) * (len(channels) - 1)
    o = oracle.UNet(dims, 1, 1, channels, strides, num_res_units=2)
    m = unet_amd.UNet(dims, 1, 1, channels, strides, num_res_units=2)
    table = m._layout()
    sd = o.state_dict()
    assert [e[0] for e in table] == list(sd.keys())
    assert [e[1] for e in table] == [tuple(v.shape) for v in sd.values()]
    assert list(m.state_dict().keys()) == list(sd.keys())
    # flat layout: disjoint, 16-byte aligned, inside the buffer, buckets contiguous
    spans = sorted((e[3], e[3] + e[2]) for e in table)
    assert all(a % 4 == 0 for a, _ in spans)
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
    assert spans[-1][1] <= m._nfloats
    buckets = m.gradient_buckets()
    assert len(buckets) == 4 and buckets[0][0] == 0 and sum(c for _, c in buckets) == m._nfloats
    assert all(buckets[i + 1][0] == buckets[i][0] + buckets[i][1] for i in range(3))
    # bucket 3 = the first encoder level (the last gradients of the backward pass), bucket 2 = the deeper encoder levels
    assert {b for name, _, _, _, b in table if name.startswith("model.0.")} == {3}
    assert (buckets[2][1] == 0) == (len(channels) == 2)
    for name, _, n, off, b in table:
        lo, cnt = m.gradient_buckets()[b]
        assert lo <= off and off + n <= lo + cnt


def test_same_seed_same_init_as_oracle():
    torch.manual_seed(123)
    o = oracle.UNet(**CFG, dropout=0.1)
    torch.manual_seed(123)
    m = unet_amd.UNet(**CFG, dropout=0.1)
    so, sm = o.state_dict(), m.state_dict()
    assert all(torch.equal(so[k], sm[k]) for k in so)
    m2 = unet_amd.UNet(**CFG)
    m2.load_state_dict(so)


def test_known_parameter_counts():
    assert sum(p.numel() for p in unet_amd.UNet(**CFG).parameters()) == 1624844
    big = dict(CFG, channels=(32, 64, 128, 256, 512))
    assert sum(p.numel() for p in unet_amd.UNet(**big).parameters()) == 6494716
    d3 = dict(CFG, spatial_dims=3)
    assert sum(p.numel() for p in unet_amd.UNet(**d3).parameters()) == 4805534
    assert len(list(unet_amd.UNet(**CFG).parameters())) == 63


def test_flop_accounting_matches_survey_figures():
    m = unet_amd.UNet(**CFG)
    p = _Plan(m._cfg_kwargs, 32, (512, 512), 1, True)
    assert abs(p.flops(False) / 32 / 1e9 - 6.036) < 0.002          # GF per 512x512 slice, forward
    assert 1.9 < p.flops(True) / p.flops(False) <= 2.0              # dgrad + wgrad (no input gradient)
    p8 = _Plan(m._cfg_kwargs, 8, (256, 256), 0, True)
    assert abs(p8.flops(False) / 1e9 - 12.07) < 0.01
    m3 = unet_amd.UNet(**dict(CFG, spatial_dims=3))
    p3 = _Plan(m3._cfg_kwargs, 4, (96, 96, 96), 1, True)
    assert abs(p3.flops(False) / 4 / 1e9 - 9.904) < 0.002
    # one launch per layer op; every conv of the network appears once in the forward list
    kernels = [s["kernel"] for s in p.steps(False)]
    nconv = sum(k.startswith(("igemm_kernel", "conv2d_", "c1_s2", "stencil")) for k in kernels)
    assert nconv == 23 - 4  # the 4 stride-2 residual convs share their unit0 launch
    assert p.workspace_bytes < 8 << 30


def test_constructor_errors_like_monai():
    with pytest.raises(ValueError):
        unet_amd.UNet(2, 1, 1, (16,), (2,), num_res_units=2)
    with pytest.raises(ValueError):
        unet_amd.UNet(2, 1, 1, (16, 32, 64), (2,), num_res_units=2)
    with pytest.raises(NotImplementedError):
        unet_amd.UNet(2, 1, 1, (16, 32), (2,), num_res_units=2, norm="BATCH")
    with pytest.raises(NotImplementedError):
        unet_amd.UNet(2, 3, 2, (16, 32), (2,), num_res_units=2)


def test_no_cpu_fallback():
    m = unet_amd.UNet(2, 1, 1, (16, 32), (2,), num_res_units=2)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, 1, 16, 16))
    with pytest.raises(RuntimeError, match="GPU"):
        unet_amd.DiceCELoss(sigmoid=True)(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))
    with pytest.raises(RuntimeError, match="GPU"):
        unet_amd.DiceMetric()(y_pred=torch.zeros(1, 1, 8, 8), y=torch.zeros(1, 1, 8, 8))
    with pytest.raises(RuntimeError, match="GPU"):
        unet_amd.sliding_window_inference(torch.zeros(1, 1, 8, 8), (4, 4), 4, lambda a: a)
    with pytest.raises(ValueError):
        unet_amd.DiceCELoss(sigmoid=True)(torch.zeros(1, 1, 8, 8), torch.zeros(1, 8, 8))
    with pytest.raises(AssertionError):
        unet_amd.DiceCELoss(sigmoid=True)(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 4))


def test_window_placement_matches_oracle():
    from oracle.inferer_ref import dense_patch_starts, _scan_interval
    for img, roi in [((512, 512), (256, 256)), ((96, 80), (64, 64)), ((256, 256), (256, 256)),
                     ((100, 100, 70), (64, 64, 64)), ((300, 200), (96, 128))]:
        interval = _scan_interval(img, roi, 0.25)
        assert unet_amd.window_starts(img, roi, 0.25) == dense_patch_starts(img, roi, interval)
    assert len(unet_amd.window_starts((512, 512), (256, 256))) == 9


def test_volume_grouping_and_checkpoint_names():
    names = ["liver_3_10.nii.gz", "liver_3_11.nii.gz", "liver_7_2.nii.gz", "liver_7_3.nii.gz"]
    d = unet_amd.per_volume_dice(names, [0.5, 1.0, float("nan"), 0.8])
    assert d == {"liver_3": 0.75, "liver_7": 0.8}


def test_checkpoint_names_follow_the_reference_branch_for_branch():
    """reference/utils/utils.py:172-187 with the flags of reference/config.py; expected strings worked out by hand
    from that function."""
    cn = unet_amd.checkpoint_name
    table = [
        (dict(organ="Task03_Liver"), "unet_task03_liver_all_2D.pth"),
        (dict(organ="Task03_Liver", train_data="all", use_pseudo_labels=True, sam_prompt="box"),
         "unet_task03_liver_box_pseudolabels_2D.pth"),                       # the README's example file (README.md:28)
        (dict(organ="Task08_HepaticVessel", use_pseudo_labels=True, sam_prompt="box_and_then_point"),
         "unet_task08_hepaticvessel_box_and_then_point_pseudolabels_2D.pth"),
        (dict(organ="Task03_Liver", train_data="n_random", n_train_samples=1000), "unet_task03_liver_1000_random_2D.pth"),
        (dict(organ="Task09_Spleen", train_data="n_worst", n_train_samples=250), "unet_task09_spleen_250_worst_2D.pth"),
        # a subset run ignores the pseudo-label flag in the name (the reference tests TRAIN_DATA first)
        (dict(organ="Task03_Liver", train_data="n_worst", use_pseudo_labels=True), "unet_task03_liver_1000_worst_2D.pth"),
        (dict(organ="Task03_Liver", img_format="3d"), "unet_task03_liver_3D.pth"),
        (dict(organ="Task03_Liver", img_format="3d", train_data="n_random", use_pseudo_labels=True), "unet_task03_liver_3D.pth"),
    ]
    for kw, want in table:
        assert cn(**kw) == want, (kw, cn(**kw))
    with pytest.raises(ValueError):
        cn("Task03_Liver", img_format="4d")


def test_unet_config_is_read_off_a_state_dict():
    from unet_amd.utils import unet_config_from_state_dict
    for dims, ch, ru in [(2, (32, 64, 128, 256, 512), 2), (2, (16, 32, 64), 1), (3, (16, 32, 64, 128, 256), 2)]:
        o = oracle.UNet(dims, 1, 1, ch, (2,) * (len(ch) - 1), num_res_units=ru)
        cfg = unet_config_from_state_dict(o.state_dict())
        assert cfg == dict(spatial_dims=dims, channels=ch, strides=(2,) * (len(ch) - 1), num_res_units=ru)
    with pytest.raises(ValueError):
        unet_config_from_state_dict({"weight": torch.zeros(1)})


# ---- LDS image of the asynchronously staged 2-D convolution (unet_amd/csrc/conv2d.hip) -------------------------
# ds_read_b128 serves a wave in four groups of 16 lanes; a group is conflict-free when its 16 lanes touch 16
# different 16-byte slots of the 256-byte bank row.
_B128_GROUPS = (
    (0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27),
    (4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31),
    (32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59),
    (36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63),
)


def _swz(x):
    return ((x >> 2) & 1) << 1


def _max_ways(addr_of_lane):
    worst = 1
    for grp in _B128_GROUPS:
        slots = [(addr_of_lane(l) // 16) % 16 for l in grp]
        worst = max(worst, max(slots.count(s) for s in set(slots)))
    return worst


@pytest.mark.parametrize("pd", [16, 17, 18])
def test_conv2d_patch_and_weight_fragment_reads_are_bank_conflict_free(pd):
    # patch fragment of lane (lr, g): pixel row `row`, column lr + dx, 16-byte chunk g of the 64-byte pixel row,
    # stored in slot g ^ swz(column); weight fragment: row lr of a 16-row block, slot g ^ swz(lr)
    ks = {16: 1, 17: 2, 18: 3}[pd]       # tap columns read from a patch of that width
    for row in range(pd):
        for dx in range(ks):
            def addr(lane, row=row, dx=dx):
                lr, g = lane & 15, lane >> 4
                x = lr + dx
                return (row * pd + x) * 64 + ((g ^ _swz(x)) << 4)
            assert _max_ways(addr) == 1, (pd, row, dx)
    for block in range(36):
        def waddr(lane, block=block):
            lr, g = lane & 15, lane >> 4
            return (block * 16 + lr) * 64 + ((g ^ _swz(lr)) << 4)
        assert _max_ways(waddr) == 1
    # without the permutation the same reads are two-way conflicts
    def plain(lane):
        lr, g = lane & 15, lane >> 4
        return lr * 64 + (g << 4)
    assert _max_ways(plain) == 2


def test_conv2d_dma_fill_is_the_inverse_of_the_fragment_permutation():
    # the DMA writes LDS lane-linearly (lane l of instruction i lands at i*1024 + 16*l), so lane (pixel = l >> 2,
    # slot = l & 3) must FETCH source chunk slot ^ swz(column); a fragment read of chunk g then finds it in slot
    # g ^ swz(column)
    pd = 18
    for pix in range(pd * pd):
        x = pix % pd
        for slot in range(4):
            src_chunk = slot ^ _swz(x)
            assert (src_chunk ^ _swz(x)) == slot
            assert 0 <= src_chunk < 4


def test_conv2d_k16_fragment_reads_are_bank_conflict_free():
    # 32-byte pixel rows, unpermuted; lanes 0..31 read tap A's pixel, lanes 32..63 tap B's (or a zero weight row)
    pd = 18
    for step in range(5):
        for wrow in range(0, 16, 4):
            for mf in range(4):
                def addr(lane, step=step, wrow=wrow, mf=mf):
                    lr, g = lane & 15, lane >> 4
                    t = 2 * step + (g >> 1)
                    if t >= 9:
                        return 22 * 1024 + (9 * 16 + lr) * 32 + (g & 1) * 16
                    return ((wrow + mf + t // 3) * pd + lr + t % 3) * 32 + (g & 1) * 16
                assert _max_ways(addr) == 1, (step, wrow, mf)
    for t in range(10):
        def waddr(lane, t=t):
            lr, g = lane & 15, lane >> 4
            return (t * 16 + lr) * 32 + (g & 1) * 16
        assert _max_ways(waddr) == 1


def _tr_b64_ways(addr_of_lane):
    """ds_read_b64_tr_b16: 8 bytes per lane, bank = (addr / 4) % 64, the two 32-lane halves are served separately."""
    worst = 1
    for half in (range(0, 32), range(32, 64)):
        banks = {}
        for l in half:
            a = addr_of_lane(l)
            for b in ((a // 4) % 64, (a // 4 + 1) % 64):
                banks.setdefault(b, set()).add(a)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def test_wide_weight_gradient_transposed_reads_are_bank_conflict_free():
    # unet_amd/csrc/wgrad3.hip: 128-byte pixel rows, 32-byte granule j of column x stored in slot j ^ ((x >> 1) & 3)
    gswz = lambda x: (x >> 1) & 3
    def pixel(lane, h):
        lr, g = lane & 15, lane >> 4
        return 16 * (g >> 1) + 8 * h + 4 * (g & 1) + (lr >> 2)
    for h in range(2):
        for blk in range(4):                                   # channel granule of the fragment
            def faddr(lane, h=h, blk=blk):
                v = pixel(lane, h)
                return v * 128 + ((blk ^ gswz(v & 15)) * 32) + (lane & 3) * 8
            assert _tr_b64_ways(faddr) == 1
            for dy in range(3):
                for dx in range(3):
                    def saddr(lane, h=h, blk=blk, dy=dy, dx=dx):
                        v = pixel(lane, h)
                        r, c = v >> 4, v & 15
                        return ((r + dy) * 18 + c + dx) * 128 + ((blk ^ gswz(c + dx)) * 32) + (lane & 3) * 8
                    assert _tr_b64_ways(saddr) == 1, (h, blk, dy, dx)
    # unpermuted rows: every second pixel row starts on the same banks
    def plain(lane):
        return pixel(lane, 0) * 128 + (lane & 3) * 8
    assert _tr_b64_ways(plain) == 4


def test_thin_weight_gradient_transposed_reads_are_bank_conflict_free():
    # unet_amd/csrc/wgrad3.hip, wgrad_thin_kernel: 64-byte pixel rows, 32-byte granule b of column x stored in slot
    # b ^ ((x >> 2) & 1); 32-byte rows are stored as they are
    def pixel(lane, h):
        lr, g = lane & 15, lane >> 4
        return 16 * (g >> 1) + 8 * h + 4 * (g & 1) + (lr >> 2)
    for rb, sw in ((64, lambda x: (x >> 2) & 1), (32, lambda x: 0)):
        for h in range(2):
            for blk in range(rb // 32):
                def faddr(lane, h=h, blk=blk):
                    v = pixel(lane, h)
                    return v * rb + ((blk ^ sw(v & 15)) * 32) + (lane & 3) * 8
                assert _tr_b64_ways(faddr) == 1
                for dy in range(3):
                    for dx in range(3):
                        def saddr(lane, h=h, blk=blk, dy=dy, dx=dx):
                            v = pixel(lane, h)
                            r, c = v >> 4, v & 15
                            return ((r + dy) * 18 + c + dx) * rb + ((blk ^ sw(c + dx)) * 32) + (lane & 3) * 8
                        assert _tr_b64_ways(saddr) == 1, (rb, h, blk, dy, dx)
    # 64-byte rows unpermuted: pixels four apart share their banks
    assert _tr_b64_ways(lambda lane: pixel(lane, 0) * 64 + (lane & 3) * 8) == 2


@pytest.mark.parametrize("shape", [(3, 32, 32), (2, 40, 56), (1, 16, 16), (1, 130, 258), (2, 34, 30)])
@pytest.mark.parametrize("rows_groups", [(2, 16), (4, 8)])
def test_strip_convolution_addresses_stay_inside_the_tensors(shape, rows_groups):
    # unet_amd/csrc/small.hip, c1_s2_fwd32_kernel: its loads are unconditional (clamped row, clamped left neighbour,
    # column 0 for lanes past the last column), so every index it forms must lie inside the image; every output pixel
    # is stored exactly once
    nb, H, W = shape
    ROWS, NG = rows_groups
    cd1, cd2 = (H + 1) // 2, (W + 1) // 2
    assert W >= 2 * cd2                         # the launcher's condition for this kernel
    lo, hi, stores = 10 ** 9, -1, set()
    for img in range(nb):
        for by in range((cd1 + ROWS * NG - 1) // (ROWS * NG)):
            for bx in range((cd2 + 63) // 64):
                for pl in range(64):
                    cx = bx * 64 + pl
                    ok = cx < cd2
                    cxs = cx if ok else 0
                    xl = 0 if cxs == 0 else -1
                    cyb = by * ROWS * NG
                    fys = [2 * cyb - 1 + j for j in range(2 * ROWS + 1)]
                    for gi in range(NG):
                        cy0 = cyb + gi * ROWS
                        if cy0 >= cd1:
                            break
                        if gi + 1 < NG:
                            fys += [2 * cy0 + 2 * ROWS + j for j in range(2 * ROWS)]
                        for r in range(ROWS):
                            if ok and cy0 + r < cd1:
                                gp = img * cd1 * cd2 + (cy0 + r) * cd2 + cx
                                assert gp not in stores
                                stores.add(gp)
                    for fy in fys:
                        fyc = min(max(fy, 0), H - 1)
                        for d in (xl, 0, 1):
                            idx = img * H * W + fyc * W + 2 * cxs + d
                            lo, hi = min(lo, idx), max(hi, idx)
    assert lo >= 0 and hi < nb * H * W
    assert len(stores) == nb * cd1 * cd2


def test_optimizer_parameter_list_is_checked_once_and_reused():
    """unet_amd.AdamW hands its own parameter list to the flat-buffer check instead of walking the module tree
    every step; the list must be exactly model.parameters() (same objects, same order), which is verified when
    it is first used for a model."""
    m = unet_amd.UNet(**dict(CFG, channels=(16, 32, 64), strides=(2, 2)))
    opt = unet_amd.AdamW(m.parameters(), lr=1e-3, model=m)
    params = opt._params_of(m)
    assert len(params) == len(list(m.parameters())) and all(a is b for a, b in zip(params, m.parameters()))
    assert opt._params_of(m) is params                      # second use: no walk, same list
    flat = m._ensure_flat(torch.device("cpu"), params)       # parameters become views of one buffer, in engine order
    assert all(p.data_ptr() >= flat.data_ptr() and p.data_ptr() < flat.data_ptr() + 4 * flat.numel() for p in params)
    other = unet_amd.UNet(**dict(CFG, channels=(16, 32, 64), strides=(2, 2)))
    with pytest.raises(RuntimeError):
        unet_amd.AdamW(list(m.parameters())[:-1], lr=1e-3, model=m)._params_of(m)
    with pytest.raises(RuntimeError):
        opt.bind(other)._params_of(other)                    # another model's parameters


def test_one_channel_operator_arguments_are_checked_on_the_host():
    """The entry points of the two layers with one channel on the fine side refuse bad geometry before any launch
    (no GPU involved): the workspace size is 0 and the call returns an error with a message."""
    lib = _lib.load()
    dims3 = lambda *v: (ctypes.c_int32 * 3)(*v)
    # [slabs][27 taps + bias][32 channels] fp32 in 3-D, [slabs][9 + 1][32] in 2-D
    assert lib.mi_conv1_wgrad_workspace_bytes(3, 32, dims3(96, 96, 96)) >= 768 * 28 * 32 * 4
    assert lib.mi_conv1_wgrad_workspace_bytes(2, 32, dims3(1, 512, 512)) % 4 == 0
    assert lib.mi_conv1_wgrad_workspace_bytes(2, 32, dims3(2, 512, 512)) == 0      # 2-D: depth must be 1
    assert lib.mi_conv1_wgrad_workspace_bytes(3, 32, dims3(96, 95, 96)) == 0       # odd size under stride 2
    assert lib.mi_conv1_wgrad_workspace_bytes(3, 12, dims3(96, 96, 96)) == 0       # channels not a multiple of 8
    assert lib.mi_conv1_wgrad_workspace_bytes(4, 32, dims3(96, 96, 96)) == 0
    t = _lib.MiTensor()
    t.ptr, t.pitch, t.coff = 256, 32, 0      # never dereferenced: every call below fails in its checks
    assert lib.mi_conv1_forward(0, 3, 1, dims3(8, 7, 8), 32, 256, 256, None, ctypes.byref(t), None, None) != 0
    assert b"odd size" in lib.mi_last_error()
    assert lib.mi_conv1_transposed(2, 3, 1, dims3(8, 8, 8), 32, ctypes.byref(t), 256, None, 256, None, None) != 0
    assert b"dtype" in lib.mi_last_error()
    assert lib.mi_conv1_wgrad(0, 3, 1, dims3(8, 8, 8), 32, None, ctypes.byref(t), 256, None, 256, None) != 0
    assert b"null" in lib.mi_last_error()
