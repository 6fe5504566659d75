"""GPU tier: every C-ABI kernel of libdpc_hip.so on a real MI355X against torch-CPU
expectations (same cases as the simulator tier, production-like tile counts).

The last element of every convolution shape is the kernel that case is MEANT to cover in bf16 mode (the name dpc_last_kernel
reports after the call, include/dpc_hip.h; "wgrad2_kernel|padded=1" = wgrad2 on a padded power-of-two grid): a size threshold
or switch that demotes the shape to another kernel fails the case even though the numbers still agree."""
import pytest
import torch

import kcases as kc
from dpc_amd import _lib as L

pytestmark = pytest.mark.gpu
F32, BF16 = torch.float32, torch.bfloat16


@pytest.fixture(scope="module")
def k():
    assert torch.cuda.is_available(), "GPU tier needs an MI355X"
    return kc.K(L.load_hip(), "cuda:0")


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape", [
    (4, 64, 64, 3, 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), "conv_halo_ws_kernel<false,8,128>"),  # layer1
    (4, 64, 128, 3, 32, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), "igemm_kernel<T,TO,BN,1>"),  # layer2.0.conv1
    (6, 128, 256, 5, 16, 16, (3, 3, 3), (2, 2, 2), (1, 1, 1), "igemm_kernel<T,TO,BN,1>"),  # layer3.0.conv1
    (16, 256, 256, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), "igemm_kernel<T,TO,BN,1>"),  # layer3 body
    (16, 256, 256, 2, 4, 4, (3, 3, 3), (1, 1, 1), (1, 1, 1), "igemm_kernel<T,TO,BN,1>"),  # layer4 body
    (5, 128, 256, 5, 16, 16, (1, 1, 1), (2, 2, 2), (0, 0, 0), "igemm_kernel<T,TO,BN,1>"),  # downsample
    (3, 16, 40, 2, 9, 7, (1, 3, 3), (1, 1, 1), (0, 1, 1), "igemm_kernel<T,TO,BN,2>"),  # ragged everything
    (70, 64, 64, 1, 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), "conv_halo_ws_kernel<false,8,128>"),  # 560 patch tiles > 256 workgroups: tile pipeline of the role-specialised kernel
    (70, 128, 128, 1, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), "igemm_wsp_kernel<false>"),  # >= 16384 rows: loader/compute specialised kernel, 70 tiles
    (90, 256, 264, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), "igemm_ws_kernel<false>"),  # same, 3x3x3, three column tiles (last ragged), ragged last row tile
    (150, 128, 128, 4, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), "igemm_wsp_kernel<false>"),  # plane variant (16 x 16 planes as patches): 600 tiles, 2-3 per workgroup
    (33, 64, 264, 2, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), "igemm_wsp_kernel<false>"),  # plane variant: one channel group, three column tiles (last ragged)
    (530, 128, 128, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), "igemm_ws_kernel<false>"),
    (110, 256, 256, 3, 14, 14, (3, 3, 3), (1, 1, 1), (1, 1, 1), "igemm_ws_kernel<false>"),  # layer3 of the 224-pixel family: 196-pixel planes, grouped tiles straddle clips
    (300, 256, 256, 2, 7, 7, (3, 3, 3), (1, 1, 1), (1, 1, 1), "igemm_ws_kernel<false>"),  # layer4 of the 224-pixel family: 49-pixel planes, T = 2   # 3x3x3 over 8 x 8 planes (layer3): temporally grouped tiles, 399 of them, ragged last clip group
])
def test_conv_fwd(k, dtype, shape):
    kc.case_conv_fwd(k, dtype, *shape[:9], expect=shape[9] if dtype == BF16 else None)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape", [
    (24, 128, 256, 3, 16, 16, (3, 3, 3), (2, 2, 2), (1, 1, 1), "igemm_ws_kernel<false,true>"),   # layer3.0.conv1 at 128^2: 12 tiles per class (8 interleaved + 4)
    (130, 256, 256, 2, 8, 8, (3, 3, 3), (2, 2, 2), (1, 1, 1), "igemm_ws_kernel<false,true>"),    # layer4.0.conv1: T = 2, one frame per temporal class
    (40, 128, 256, 3, 28, 28, (3, 3, 3), (2, 2, 2), (1, 1, 1), "igemm_ws_kernel<false,true>"),   # layer3.0.conv1 of the 224-pixel family
    (8, 128, 128, 2, 32, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), "igemm_ws_kernel<false,true>"),    # 2D stride, two channel groups
    (24, 128, 256, 3, 15, 17, (3, 3, 3), (2, 2, 2), (1, 1, 1), "igemm_ws_kernel<false,true>"),    # odd extents: unequal classes, class after class
    (64, 64, 128, 1, 32, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), "igemm_kernel<T,TO,BN,3>"),        # layer2.0.conv1 (64 output columns), few planes: generic kernel, interleaved classes
    (600, 64, 128, 1, 32, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), "igemm_wsd_kernel"),              # layer2.0.conv1 at 128^2 (round 6): 2 x 2 shift convolution over 16 x 16 gradient planes
    (131, 64, 128, 4, 32, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), "igemm_wsd_kernel"),              # 524 planes: kinds A / B walk 3-4 / 4-5 planes, ragged ends
    (40, 64, 128, 1, 56, 56, (1, 3, 3), (1, 2, 2), (0, 1, 1), "igemm_kernel<T,TO,BN,3>"),        # the same layer of the 224-pixel family (classes rotate over the rounds)
])
def test_conv_dgrad_strided(k, dtype, shape):
    """strided input-gradients without a residual (a layer's first conv): parity classes on the loader / compute kernel"""
    kc.case_conv_dgrad(k, dtype, *shape[:9], expect=shape[9] if dtype == BF16 else None, with_add=False)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape", [
    (4, 64, 64, 3, 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), "conv_halo_ws_kernel<true,8,128>"),
    (4, 64, 128, 3, 32, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), "igemm_kernel<T,TO,BN,3>"),
    (6, 128, 256, 5, 16, 16, (3, 3, 3), (2, 2, 2), (1, 1, 1), "igemm_kernel<T,TO,BN,3>"),
    (16, 256, 256, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), "igemm_kernel<T,TO,BN,1>"),
    (5, 128, 256, 5, 16, 16, (1, 1, 1), (2, 2, 2), (0, 0, 0), "igemm_kernel<T,TO,BN,3>"),
    (66, 64, 64, 1, 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), "conv_halo_ws_kernel<true,8,128>"),  # patch kernel, flipped taps, 528 tiles
    (66, 128, 128, 1, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), "igemm_wsp_kernel<true>"),  # unit-stride input-gradient on the specialised kernel
    (87, 256, 256, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), "igemm_ws_kernel<true>"),
    (3, 16, 32, 2, 9, 7, (1, 3, 3), (1, 2, 2), (0, 1, 1), "igemm_kernel<T,TO,BN,0>"),
    (150, 128, 128, 4, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), "igemm_wsp_kernel<true>"),  # plane variant, flipped taps + residual addend, 600 tiles
    (40, 128, 256, 2, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), "igemm_wsp_kernel<true>"),  # plane variant, four channel groups
    (300, 256, 256, 2, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), "igemm_ws_kernel<true>"),  # same family, T = 2, flipped taps + residual addend
    (110, 256, 256, 3, 14, 14, (3, 3, 3), (1, 1, 1), (1, 1, 1), "igemm_ws_kernel<true>"),  # 224-pixel family, flipped taps + residual addend
])
def test_conv_dgrad(k, dtype, shape):
    kc.case_conv_dgrad(k, dtype, *shape[:9], expect=shape[9] if dtype == BF16 else None)


@pytest.mark.parametrize("shape,gate", [
    ((66, 64, 64, 1, 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), "conv_halo_ws_kernel<true,8,128>"), False),
    ((66, 64, 64, 1, 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), "conv_halo_ws_kernel<true,8,128,true>"), True),
    ((66, 128, 128, 1, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), "igemm_wsp_kernel<true>"), False),
    ((87, 256, 256, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), "igemm_ws_kernel<true>"), False),
    ((16, 256, 256, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), "igemm_kernel<T,TO,BN,1>"), False),
    ((16, 256, 256, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), "igemm_kernel<T,TO,BN,1,true>"), True),   # the engine's last block: gated, generic kernel
    ((6, 128, 256, 5, 16, 16, (3, 3, 3), (2, 2, 2), (1, 1, 1), "igemm_kernel<T,TO,BN,3>"), False),      # strided: parity classes, in-place accumulation
    ((5, 128, 256, 5, 16, 16, (1, 1, 1), (2, 2, 2), (0, 0, 0), "igemm_kernel<T,TO,BN,3>"), False),      # the in-place 1x1 downsample gradient
])
def test_conv_dgrad_addend_may_alias_out(k, shape, gate):
    """include/dpc_hip.h: addend == out is part of the contract -- every dispatch variant that takes an addend, both ways"""
    kc.case_conv_dgrad_alias(k, BF16, *shape[:9], expect=shape[9], gate=gate)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape", [
    (4, 64, 64, 3, 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), "wgrad_patch_kernel<32>"),
    (4, 64, 128, 3, 32, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), "wgrad2_kernel|padded=0"),
    (6, 128, 256, 5, 16, 16, (3, 3, 3), (2, 2, 2), (1, 1, 1), "wgrad2_kernel|padded=0"),
    (16, 256, 256, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), "wgrad_patch_kernel<8>"),
    (5, 128, 256, 5, 16, 16, (1, 1, 1), (2, 2, 2), (0, 0, 0), "wgrad2_kernel|padded=0"),
    (5, 8, 24, 2, 6, 6, (1, 3, 3), (1, 2, 2), (0, 1, 1), "wgrad_kernel<T,64,64,RF>"),
    (6, 64, 64, 2, 56, 56, (1, 3, 3), (1, 1, 1), (0, 1, 1), "wgrad_patch_kernel<64>"),  # 224-pixel family: staged-patch kernel with padded widths
    (6, 128, 128, 2, 28, 28, (1, 3, 3), (1, 1, 1), (0, 1, 1), "wgrad_patch_kernel<32>"),
    (8, 256, 256, 3, 14, 14, (3, 3, 3), (1, 1, 1), (1, 1, 1), "wgrad_patch_kernel<16>"),
    (16, 256, 256, 2, 7, 7, (3, 3, 3), (1, 1, 1), (1, 1, 1), "wgrad_patch_kernel<8>"),
    (6, 64, 128, 2, 56, 56, (1, 3, 3), (1, 2, 2), (0, 1, 1), "wgrad2_kernel|padded=1"),  # 224-pixel family, strided: transpose-read kernel on padded grids (28 -> 32)
    (8, 128, 256, 5, 28, 28, (3, 3, 3), (2, 2, 2), (1, 1, 1), "wgrad2_kernel|padded=1"),  # 14 x 14 output padded to 16 x 16
    (16, 256, 256, 3, 14, 14, (3, 3, 3), (2, 2, 2), (1, 1, 1), "wgrad2_kernel|padded=1"),  # 7 x 7 output padded to 8 x 8: one plane per chunk
    (8, 128, 256, 5, 28, 28, (1, 1, 1), (2, 2, 2), (0, 0, 0), "wgrad2_kernel|padded=1"),  # strided 1x1 downsample
])
def test_conv_wgrad(k, dtype, shape):
    kc.case_conv_wgrad(k, dtype, *shape[:9], expect=shape[9] if dtype == BF16 else None)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape,gate,bn_relu,kern", [
    ((66, 64, 64, 1, 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1)), True, True, "conv_halo_ws_kernel<%s,8,128,true>"),   # layer1, 528 tiles
    ((3, 64, 64, 2, 56, 56, (1, 3, 3), (1, 1, 1), (0, 1, 1)), True, False, "conv_halo_ws_kernel<%s,8,128,true>"),   # 224-pixel family, ragged tiles
    ((16, 256, 256, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1)), True, True, "igemm_kernel<T,TO,BN,1,true>"),         # generic kernel
    ((66, 128, 128, 1, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1)), True, True, None),
    ((87, 256, 256, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1)), True, True, None),
])
def test_conv_dgrad_ex(k, dtype, shape, gate, bn_relu, kern):
    """dpc_conv_igemm_ex: gated residual addend + fused BatchNorm-backward sums (residual form and the plain form of conv2)"""
    ex = lambda add: (kern % add if "%s" in kern else kern) if (kern and dtype == BF16) else None  # noqa: E731
    kc.case_conv_dgrad_ex(k, dtype, *shape, gate=gate, bnred=True, bn_relu=bn_relu, expect=ex("true"))
    kc.case_conv_dgrad_ex(k, dtype, *shape, bnred=True, bn_relu=bn_relu, with_add=False, expect=ex("false"))
    kc.case_conv_dgrad_ex(k, dtype, *shape, gate=True, bnred=False, expect=ex("true"))


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("mnk", [(6144, 6144, 256), (6468, 6468, 256), (15680, 15680, 256), (2048, 768, 256), (192, 192, 256), (130, 70, 64)])
def test_gemm_nt(k, dtype, mnk):
    big = dtype == BF16 and mnk[2] == 256 and mnk[0] >= 1024 and mnk[1] >= 512   # the materialised score: dedicated kernel
    v2 = mnk[0] * mnk[1] >= (1 << 26) and mnk[1] % 32 == 0   # large outputs with cache-line-aligned rows: the 8-wave form
    kc.case_gemm_nt(k, dtype, *mnk, expect=("score_gemm2_kernel<16>" if v2 else "score_gemm_kernel<16>") if big else ("igemm_kernel" if dtype == BF16 else None))


def test_score_gemm_bf16_logits(k):
    """the train step's logits in the compute dtype (round 6): cfg2 and cfg5 size, and a ragged row count"""
    for mnk in ((6144, 6144, 256), (15680, 15680, 256), (1100, 1088, 256)):
        kc.case_gemm_nt_bf16out(k, *mnk, expect="score_gemm2_kernel<16,true>")


def test_score_gemm_8wave_form_on_ragged_shapes(k, monkeypatch):
    """score_gemm2_kernel forced onto shapes it does not serve by default: ragged last row block (waves without any row), rows that
    are not cache-line multiples, R = 6 144"""
    monkeypatch.setenv("DPC_SCORE_GEMM2", "2")
    for mnk in ((6468, 6468, 256), (6144, 6144, 256), (1100, 1028, 256)):
        kc.case_gemm_nt(k, BF16, *mnk, expect="score_gemm2_kernel<16>")


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("mnk", [(6144, 256, 6144), (6468, 256, 6472), (15680, 256, 15680), (2048, 256, 256)])
def test_gemm_nt_splitk(k, dtype, mnk):
    """d_pred = dS @ feature_inf at cfg2 / cfg4 (leading dimension 6 472 = R rounded up to the 16-byte unit) / cfg5 size:
    reduction split over workgroups, f32 slabs"""
    ws = "gemm_ws_kernel<false>" if dtype == BF16 and mnk[2] >= 1024 else None   # long reductions: loader / compute kernel
    ns = kc.case_gemm_nt_splitk(k, dtype, *mnk, pad=8, expect=ws)
    assert ns >= (2 if mnk[0] > 4096 else 1)


@pytest.mark.parametrize("mnk,pad", [((6144, 256, 6144), 8), ((6468, 256, 6468), 4), ((15680, 256, 15680), 8), ((1024, 128, 1032), 8)])
def test_gemm_tn_splitk(k, mnk, pad):
    """d_feature_inf = dS^T @ pred at cfg2 / cfg4 (R = 6 468: rows of 6 472 elements, the last 16-byte unit of every operand row
    straddles the end of the operand) / cfg5 size: A read K-major through transpose reads, f32 slabs"""
    ns = kc.case_gemm_tn_splitk(k, *mnk, pad=pad)
    assert ns >= (2 if mnk[0] > 4096 else 1)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_stem_s2d(k, dtype):
    ws = "conv_halo_ws_kernel<false,2,256>" if dtype == BF16 else None
    kc.case_stem(k, dtype, 4, 5, 64, 64, expect=(ws, ws and "wgrad2_kernel|padded=0"))
    kc.case_stem(k, dtype, 2, 2, 16, 20, expect=(ws, ws and "wgrad2_kernel|padded=1"))
    kc.case_stem(k, dtype, 3, 2, 128, 128, expect=(ws, ws and "wgrad_stem_kernel"))  # the real stem geometry: staged-patch weight gradient (bf16)
    kc.case_stem(k, dtype, 1, 2, 32, 224, expect=(ws, ws and "wgrad_stem_kernel"))   # two 64-column segments per row


@pytest.mark.parametrize("rows,C,misalign", [(1, 64, 0), (3, 5, 0), (37, 70, 0), (131, 64, 0), (300, 128, 0), (1030, 64, 0), (256, 512, 0), (129, 64, 1)])
def test_bn_finalize(k, rows, C, misalign):
    """the compact finalize kernels (256 threads: they must start beside a side-stream weight gradient that holds every CU) against f64 sums"""
    kc.case_bn_finalize(k, rows, C, misalign)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("relu,res_mode,C", [(True, 0, 64), (True, 1, 128), (False, 2, 256), (True, 2, 256), (True, 1, 16)])
def test_bn(k, dtype, relu, res_mode, C):
    kc.case_bn_fwd_bwd(k, dtype, 20011, C, relu, res_mode)


@pytest.mark.parametrize("unroll", ["4", "1"])
@pytest.mark.parametrize("relu,res_mode,C", [(True, 0, 64), (True, 1, 128), (False, 2, 256)])
def test_bn_streaming_forms(k, monkeypatch, unroll, relu, res_mode, C):
    """the >= 192 MB instantiations of bn_apply / bn_bwd_apply (non-temporal, U units per thread in flight) on a 200 MB tensor"""
    monkeypatch.setenv("DPC_BN_UNROLL", unroll)
    kc.case_bn_fwd_bwd(k, BF16, (200 << 20) // (2 * C) + 37, C, relu, res_mode)


@pytest.mark.parametrize("unroll", ["4", "1"])
@pytest.mark.parametrize("rows,C,relu,res_mode", [(196608 + 5, 256, True, 1), (32768 + 3, 256, True, 0), (40000, 128, False, 2)])
def test_bn_small_tensor_forms(k, monkeypatch, unroll, rows, C, relu, res_mode):
    """the instantiations for tensors below the streaming threshold at layer3 / layer4 size of cfg2 (U units per thread in flight)"""
    monkeypatch.setenv("DPC_BN_SMALL_UNROLL", unroll)
    kc.case_bn_fwd_bwd(k, BF16, rows, C, relu, res_mode)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("hw", [(64, 64), (7, 10)])
def test_stem_pool(k, dtype, hw):
    kc.case_stem_pool(k, dtype, 6, hw[0], hw[1], 64)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_tpool_split(k, dtype):
    kc.case_tpool_split(k, dtype, 4, 8, 2, 16, 256, 3)
    kc.case_tpool_split(k, dtype, 3, 8, 2, 49, 256, 5)


def test_reduce_unpack_forms(k):
    """every kernel behind dpc_reduce_unpack"""
    kc.case_reduce_unpack(k, 5, 6144, 1, 256, False, False, expect="reduce_unpack_few_kernel")    # the score backward's reductions at cfg2
    kc.case_reduce_unpack(k, 2, 15680, 1, 256, False, True, expect="reduce_unpack_few_kernel")    # cfg5, accumulate
    kc.case_reduce_unpack(k, 8, 2049, 1, 128, True, False, expect="reduce_unpack_few_kernel")     # permuted strides
    kc.case_reduce_unpack(k, 300, 64, 9, 64, True, True, expect="reduce_unpack4_kernel")       # many slabs: the split-lane form
    kc.case_reduce_unpack(k, 512, 128, 1, 64, False, False, expect="reduce_unpack4_kernel")    # layer2.0 downsample gradient at cfg2: 512 slabs x 32 KB
    kc.case_reduce_unpack(k, 4, 256, 27, 256, True, False, expect="reduce_unpack_t_kernel")    # conv layout, few slabs
    kc.case_reduce_unpack(k, 3, 33, 5, 7, False, True, expect="reduce_unpack_kernel")          # unaligned: element form


@pytest.mark.parametrize("bps", [(4, 3, 16), (2, 1, 4), (3, 5, 49), (16, 3, 16)])
def test_mask(k, bps):
    kc.case_mask(k, *bps)


@pytest.mark.parametrize("dtype_d", [F32, BF16])
def test_ce_topk(k, dtype_d):
    kc.case_ce_topk(k, 24, 24, dtype_d)
    kc.case_ce_topk(k, 1764, 1764, dtype_d)
    kc.case_ce_topk(k, 6144, 6144, dtype_d)
    kc.case_ce_topk(k, 15680, 15680, dtype_d)  # cfg5: 983 MB of logits


def test_ce_topk_bf16_logits(k):
    for r in (24, 1768, 6144, 15680):
        kc.case_ce_topk_bf16(k, r, r)


def test_adam(k):
    kc.case_adam(k, 1027)
    kc.case_adam(k, 14583104)


def test_transpose(k):
    kc.case_transpose(k, 6468, 256)
    kc.case_transpose_x2(k, 6468, 256)
    kc.case_transpose_x2(k, 15680, 256)


def test_copy2d_multi(k):
    kc.case_copy2d_multi(k)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_pack3d_multi(k, dtype):
    kc.case_pack3d_multi(k, dtype, [(64, 64, 9), (128, 64, 9), (256, 128, 27), (512, 256, 27), (256, 128, 1), (70, 24, 27)])


def test_dropout_mask(k):
    kc.case_dropout_mask(k, 1027, 0.1, 233, 0)
    kc.case_dropout_mask(k, 7 * 2048 * 256, 0.1, (5 << 32) | 77, 12)  # cfg2: masks of all 7 recurrence steps


def test_adam_dev(k):
    kc.case_adam_dev(k, 1027)
    kc.case_adam_dev(k, 14583104)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape", [(3, 16, 256, 3, 5), (2, 49, 256, 5, 3), (4, 4, 32, 3, 5), (40, 16, 256, 3, 5)])
def test_gru_chain(k, dtype, shape):
    kc.case_gru_chain(k, dtype, *shape)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_gru_chain_philox(k, dtype):
    kc.case_gru_chain_philox(k, dtype, 3, 16, 256, 3, 5)


def test_gru_chain_four_waves(k, monkeypatch):
    """the four-wave form of the D = 256 recurrence (two column tiles per wave; the default is eight waves)"""
    monkeypatch.setenv("DPC_GRU_WAVES", "4")
    kc.case_gru_chain(k, BF16, 40, 16, 256, 3, 5)
    kc.case_gru_chain_philox(k, F32, 3, 16, 256, 3, 5)


def test_gru_chain_reference_fixture(k, golden_dir):
    import os
    import numpy as np
    kc.case_gru_chain_golden(k, np.load(os.path.join(golden_dir, "ops.npz")))


@pytest.mark.parametrize("rd", [(6144, 256), (6468, 256), (192, 256), (200, 32)])
def test_score_fused(k, rd):
    kc.case_score_fused(k, *rd)


def test_score_fused_cfg5(k):
    kc.case_score_fused(k, 15680, 256, check_score=False)  # cfg5: the 983 MB matrix never exists on the device


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape", [(5, 64, 128, 3, 32, 32, (1, 1, 1), (1, 2, 2), (0, 0, 0), 'igemm_kernel<T,TO,BN,3>'), (5, 128, 256, 5, 16, 16, (1, 1, 1), (2, 2, 2), (0, 0, 0), 'igemm_kernel<T,TO,BN,3>'), (3, 64, 128, 2, 32, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), 'igemm_kernel<T,TO,BN,3>')])
def test_conv_dgrad_inplace(k, dtype, shape):
    kc.case_conv_dgrad_inplace(k, dtype, *shape[:9], expect=shape[9] if dtype == BF16 else None)


def test_stem_wgrad_fused(k):
    kc.case_stem_wgrad_fused(k, 3, 2, 128, 128)
    kc.case_stem_wgrad_fused(k, 2, 1, 64, 224)
    kc.case_stem_wgrad_fused(k, 1, 2, 30, 200)
