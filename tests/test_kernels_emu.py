"""CPU tier: the gfx950 kernel sources executed by the host SIMT simulator
(tests/simt_emu) against torch-CPU expectations.  Small shapes; covers ragged
tiles, strides, padding, both element types and every C-ABI entry point."""
import os
import subprocess

import pytest
import torch

import kcases as kc
from dpc_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32, BF16 = torch.float32, torch.bfloat16


@pytest.fixture(scope="module")
def k():
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    return kc.K(L.load_emulator(), "cpu")


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_gemm_nt_splitk(k, dtype):
    assert kc.case_gemm_nt_splitk(k, dtype, 130, 64, 1024, pad=8) > 1   # ragged rows, several K splits, padded leading dimensions
    kc.case_gemm_nt_splitk(k, dtype, 200, 72, 520)                       # ragged columns (element-wise tail path), K not a multiple of a chunk
    assert kc.case_gemm_nt_splitk(k, dtype, 64, 32, 64) == 1             # nothing to split


def test_gemm_ws_splitk(k, monkeypatch):
    """the loader / compute split-K GEMMs of the score backward (csrc/gemm_ws.hip) on small shapes: ragged last row tile, K tail inside
    a chunk, several slices, padded leading dimensions; row-major A and K-major A (transpose reads)"""
    monkeypatch.setenv("DPC_GEMM_WS_MIN", "64")
    assert kc.case_gemm_nt_splitk(k, BF16, 300, 128, 1096, pad=8, expect="gemm_ws_kernel<false>") > 1
    kc.case_gemm_nt_splitk(k, BF16, 256, 256, 72, expect="gemm_ws_kernel<false>")
    assert kc.case_gemm_tn_splitk(k, 264, 128, 1096, pad=8) > 1
    kc.case_gemm_tn_splitk(k, 520, 256, 200, pad=16)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape,gate,bnred,bn_relu", [
    ((1, 32, 32, 3, 6, 6, (3, 3, 3), (1, 1, 1), (1, 1, 1)), True, True, True),     # generic kernel, 3x3x3
    ((2, 8, 64, 2, 6, 5, (1, 3, 3), (1, 1, 1), (0, 1, 1)), True, True, True),      # bf16: role-specialised patch kernel (EPI), residual
    ((2, 64, 64, 2, 9, 33, (1, 3, 3), (1, 1, 1), (0, 1, 1)), True, True, False),   # several tiles per workgroup, no ReLU on the reduced unit
    ((2, 64, 64, 2, 9, 33, (1, 3, 3), (1, 1, 1), (0, 1, 1)), False, True, True),   # plain addend + fused reduction
    ((2, 64, 64, 1, 12, 20, (1, 3, 3), (1, 1, 1), (0, 1, 1)), True, False, False), # gated addend only
    ((1, 128, 128, 1, 10, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1)), True, True, True),
])
def test_conv_dgrad_ex(k, dtype, shape, gate, bnred, bn_relu):
    kc.case_conv_dgrad_ex(k, dtype, *shape, gate=gate, bnred=bnred, bn_relu=bn_relu)
    if bnred:  # the block's second conv: no residual, only the fused reduction
        kc.case_conv_dgrad_ex(k, dtype, *shape, bnred=True, bn_relu=bn_relu, with_add=False)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape", [
    (2, 16, 64, 2, 9, 9, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    (2, 64, 72, 3, 8, 8, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (3, 32, 160, 2, 7, 5, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    (1, 8, 24, 5, 6, 6, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 64, 64, 2, 9, 33, (1, 3, 3), (1, 1, 1), (0, 1, 1)),   # LDS-staged patch kernel (bf16: 128 B/pos, f32: 256 B/pos)
    (1, 64, 72, 1, 20, 12, (1, 3, 3), (1, 1, 1), (0, 1, 1)),  # narrow image: 8x16 tiles
    (2, 32, 64, 2, 6, 7, (1, 3, 3), (1, 1, 1), (0, 1, 1)),    # f32 only takes the patch kernel (128 B/pos)
    (1, 128, 128, 2, 10, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
])
def test_conv_fwd(k, dtype, shape):
    kc.case_conv_fwd(k, dtype, *shape)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape", [
    (2, 16, 64, 2, 9, 9, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    (2, 16, 32, 5, 8, 8, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (1, 32, 32, 3, 6, 6, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 8, 16, 2, 8, 8, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    (1, 16, 64, 3, 5, 5, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 8, 64, 2, 6, 5, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    (2, 16, 64, 5, 8, 8, (3, 3, 3), (2, 2, 2), (1, 1, 1)),      # parity classes, 8 of them
    (1, 8, 64, 5, 7, 9, (3, 3, 3), (2, 2, 2), (1, 1, 1)),       # odd extents: unequal classes
    (2, 8, 64, 2, 8, 8, (1, 1, 1), (1, 2, 2), (0, 0, 0)),       # strided 1x1x1: 3 of 4 classes see no tap
    (2, 8, 128, 3, 6, 6, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    (1, 8, 64, 1, 64, 72, (1, 3, 3), (1, 2, 2), (0, 1, 1)),     # 9 tiles per class: interleaved class order (a block of 8 + a tail)
    (1, 8, 64, 3, 32, 40, (3, 3, 3), (2, 2, 2), (1, 1, 1)),     # two temporal classes of different size, spatial classes interleaved
    (2, 64, 64, 2, 9, 33, (1, 3, 3), (1, 1, 1), (0, 1, 1)),    # patch kernel, flipped taps
    (1, 128, 128, 1, 10, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
])
def test_conv_dgrad(k, dtype, shape):
    kc.case_conv_dgrad(k, dtype, *shape)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape,gate", [
    ((2, 64, 64, 2, 9, 33, (1, 3, 3), (1, 1, 1), (0, 1, 1)), False),     # patch kernel (bf16: role-specialised)
    ((2, 64, 64, 2, 9, 33, (1, 3, 3), (1, 1, 1), (0, 1, 1)), True),      # ... with the addend gated by a ReLU mask (dpc_conv_igemm_ex)
    ((1, 32, 32, 3, 6, 6, (3, 3, 3), (1, 1, 1), (1, 1, 1)), False),      # generic kernel
    ((2, 16, 64, 5, 8, 8, (3, 3, 3), (2, 2, 2), (1, 1, 1)), False),      # strided: parity classes
    ((2, 8, 64, 2, 8, 8, (1, 1, 1), (1, 2, 2), (0, 0, 0)), False),       # strided 1x1x1: 3 of 4 classes see no tap (left as they are)
])
def test_conv_dgrad_addend_may_alias_out(k, dtype, shape, gate):
    kc.case_conv_dgrad_alias(k, dtype, *shape, gate=gate)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape", [
    (2, 16, 64, 2, 9, 9, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    (2, 64, 72, 3, 8, 8, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (3, 32, 160, 2, 7, 5, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    (5, 8, 24, 2, 6, 6, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    (2, 64, 64, 1, 8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)),      # transpose-read kernel, 1x3 waves, chunk inside a plane
    (2, 64, 136, 2, 8, 16, (1, 1, 1), (1, 2, 2), (0, 0, 0)),    # 2x2 waves, chunk spans planes (bf16)
    (1, 64, 128, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1)),     # bf16: staged-patch kernel, 3 temporal taps, W = 8; f32: 2x3 waves
    (2, 64, 128, 3, 4, 4, (3, 3, 3), (1, 1, 1), (1, 1, 1)),     # 2x3 waves, chunk spans planes
    (2, 256, 64, 1, 8, 8, (1, 1, 1), (1, 1, 1), (0, 0, 0)),     # 1x4 waves
    (3, 64, 64, 1, 8, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1)),     # staged-patch kernel (bf16), W = 32, 2 rows per chunk
    (2, 128, 64, 2, 8, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1)),    # staged-patch kernel, W = 16, two ci tiles
    (2, 64, 64, 2, 7, 7, (1, 3, 3), (1, 1, 1), (0, 1, 1)),      # staged-patch kernel, 7x7 image padded to width 8
    (1, 16, 64, 1, 2, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),     # rows longer than an f32 chunk (32 positions): general kernel
    (1, 16, 72, 1, 1, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),    # rows longer than a bf16 chunk too
    (1, 64, 128, 2, 14, 14, (3, 3, 3), (1, 1, 1), (1, 1, 1)),   # 14x14 -> width 16, ragged last chunk of a plane
    (1, 64, 64, 1, 5, 56, (1, 3, 3), (1, 1, 1), (0, 1, 1)),     # width 56 -> 64: one image row per chunk
    (2, 64, 128, 1, 28, 27, (1, 3, 3), (1, 2, 2), (0, 1, 1)),   # strided, 14 x 14 output: transpose-read kernel on a grid padded to 16 x 16
    (2, 64, 64, 3, 14, 14, (3, 3, 3), (2, 2, 2), (1, 1, 1)),    # 3x3x3 stride 2, 7 x 7 output padded to 8 x 8 (one plane per chunk)
    (3, 64, 128, 2, 14, 14, (1, 1, 1), (1, 2, 2), (0, 0, 0)),   # strided 1x1 (downsample), 7 x 7 output padded to 8 x 8
])
def test_conv_wgrad(k, dtype, shape):
    kc.case_conv_wgrad(k, dtype, *shape)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("mnk", [(200, 100, 264), (24, 24, 256), (130, 70, 64)])
def test_gemm_nt(k, dtype, mnk):
    kc.case_gemm_nt(k, dtype, *mnk)


@pytest.mark.parametrize("mn,kern", [((1100, 1028), "score_gemm2_kernel<2>"), ((1027, 1027), "score_gemm_kernel<2>"), ((1030, 1092), "score_gemm2_kernel<2>")])
def test_score_gemm(k, mn, kern, monkeypatch):
    """large bf16 -> f32 NT GEMM with a short reduction (the materialised score): register-resident rows, streamed column tiles.
    1100 x 1028 / 1030 x 1092: the 8-wave form (staged whole-row stores; ragged last row block incl. waves without any row, ragged last
    column tile); 1027: odd leading dimension -> the 4-wave form with element-wise stores"""
    monkeypatch.setenv("DPC_SCORE_GEMM2", "2")   # the 8-wave form regardless of the output size (by default it serves >= 2^26 elements)
    kc.case_gemm_nt(k, BF16, mn[0], mn[1], 32, expect=kern)


def test_score_gemm_bf16_logits(k):
    """the same product written in the compute dtype (the train step's logits, round 6): 8-wave form, 128-byte staged rows; ragged
    last row block, several column tiles per split"""
    kc.case_gemm_nt_bf16out(k, 1100, 1088, 32, expect="score_gemm2_kernel<2,true>")
    kc.case_gemm_nt_bf16out(k, 520, 2048, 32, seed=9, expect="score_gemm2_kernel<2,true>")


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_stem_s2d(k, dtype):
    kc.case_stem(k, dtype, 2, 2, 16, 20)
    kc.case_stem(k, dtype, 2, 1, 16, 64)   # every tile inside the image: partial sums taken by the compute waves (bf16)
    kc.case_stem(k, dtype, 1, 1, 4, 168)   # 84 columns: two 64-position row segments in the staged-patch weight gradient
    kc.case_stem(k, dtype, 1, 2, 6, 100)   # 50 columns after space-to-depth: staged-patch weight gradient (bf16), ragged row segment


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("relu,res_mode,C", [(True, 0, 64), (True, 1, 16), (False, 2, 48), (True, 2, 256)])
def test_bn(k, dtype, relu, res_mode, C):
    kc.case_bn_fwd_bwd(k, dtype, 333, C, relu, res_mode)


@pytest.mark.parametrize("rows,C,misalign", [(1, 64, 0), (3, 5, 0), (37, 70, 0), (131, 64, 0), (300, 128, 0), (1030, 64, 0), (256, 512, 0), (129, 64, 1)])
def test_bn_finalize(k, rows, C, misalign):
    """the compact finalize kernels (256 threads: they must start beside a side-stream weight gradient that holds every CU) against f64 sums"""
    kc.case_bn_finalize(k, rows, C, misalign)


@pytest.mark.parametrize("unroll", ["4", "2", "1"])
@pytest.mark.parametrize("relu,res_mode,C", [(True, 0, 64), (True, 1, 128), (False, 2, 256)])
def test_bn_streaming_forms(k, monkeypatch, unroll, relu, res_mode, C):
    """the large-tensor instantiations (non-temporal accesses, U units per thread in flight: DPC_BN_UNROLL) on a tensor just above
    a lowered threshold: spans that are not multiples of the unrolled stride, a ragged tail"""
    monkeypatch.setenv("DPC_BN_NT_MB", "1")
    monkeypatch.setenv("DPC_BN_UNROLL", unroll)
    monkeypatch.setenv("DPC_BN_APPLY_GRID", "7")
    kc.case_bn_fwd_bwd(k, BF16, (1 << 20) // (2 * C) + 37, C, relu, res_mode)


@pytest.mark.parametrize("unroll", ["4", "2", "1"])
@pytest.mark.parametrize("relu,res_mode,C", [(True, 0, 64), (True, 1, 256), (False, 2, 128)])
def test_bn_small_tensor_forms(k, monkeypatch, unroll, relu, res_mode, C):
    """the instantiations for tensors below the streaming threshold (default cache policy, DPC_BN_SMALL_UNROLL units per thread in
    flight) with few workgroups: several unrolled iterations per thread, spans that end inside an unrolled stride, a ragged tail"""
    monkeypatch.setenv("DPC_BN_SMALL_UNROLL", unroll)
    monkeypatch.setenv("DPC_BN_SMALL_GRID", "3")
    kc.case_bn_fwd_bwd(k, BF16, (1 << 19) // (2 * C) + 21, C, relu, res_mode)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("hw", [(8, 8), (7, 10)])
def test_stem_pool(k, dtype, hw):
    kc.case_stem_pool(k, dtype, 3, hw[0], hw[1], 16)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_tpool_split(k, dtype):
    kc.case_tpool_split(k, dtype, 2, 8, 2, 4, 32, 3)
    kc.case_tpool_split(k, dtype, 3, 5, 1, 9, 16, 2)


def test_reduce_unpack_forms(k):
    """every kernel behind dpc_reduce_unpack"""
    kc.case_reduce_unpack(k, 5, 1100, 1, 256, False, False, expect="reduce_unpack_few_kernel")    # few slabs, many sums (score backward)
    kc.case_reduce_unpack(k, 3, 300, 4, 256, False, True, expect="reduce_unpack_few_kernel")      # + accumulate
    kc.case_reduce_unpack(k, 8, 2049, 1, 128, True, False, expect="reduce_unpack_few_kernel")     # permuted strides
    kc.case_reduce_unpack(k, 19, 64, 9, 64, True, True, expect="reduce_unpack4_kernel")        # many slabs: the split-lane form
    kc.case_reduce_unpack(k, 150, 8, 1, 64, False, False, expect="reduce_unpack4_kernel")      # two eight-slab trips per split lane + a ragged tail
    kc.case_reduce_unpack(k, 64, 8, 1, 32, False, True, expect="reduce_unpack4_kernel")        # exactly one trip, empty tail
    kc.case_reduce_unpack(k, 4, 64, 9, 64, True, False, expect="reduce_unpack_t_kernel")       # conv layout, few slabs
    kc.case_reduce_unpack(k, 3, 33, 5, 7, False, True, expect="reduce_unpack_kernel")          # unaligned: element form


@pytest.mark.parametrize("bps", [(4, 3, 16), (2, 1, 4), (3, 5, 9)])
def test_mask(k, bps):
    kc.case_mask(k, *bps)


@pytest.mark.parametrize("dtype_d", [F32, BF16])
def test_ce_topk(k, dtype_d):
    kc.case_ce_topk(k, 24, 24, dtype_d)
    kc.case_ce_topk(k, 300, 300, dtype_d)
    kc.case_ce_topk(k, 37, 37, dtype_d)      # odd width: the three-sweep kernel
    kc.case_ce_topk(k, 1100, 1100, dtype_d)  # register-resident rows, NV = 8, more than one float4 per thread


def test_ce_topk_bf16_logits(k):
    kc.case_ce_topk_bf16(k, 24, 24)
    kc.case_ce_topk_bf16(k, 304, 304)      # ragged: 38 units over 256 threads
    kc.case_ce_topk_bf16(k, 2304, 2304)    # more than one unit per thread


def test_adam(k):
    kc.case_adam(k, 1027)


def test_transpose(k):
    kc.case_transpose(k, 70, 45)
    kc.case_transpose_x2(k, 132, 72)   # ragged tiles in both directions, padded output rows


def test_copy2d_multi(k):
    kc.case_copy2d_multi(k)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_pack3d_multi(k, dtype):
    kc.case_pack3d_multi(k, dtype, [(24, 40, 9), (70, 8, 27), (16, 130, 1), (8, 8, 16), (3, 5, 70)])   # ragged tiles, 1x1, taps beyond the tile


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10 pin the numpy generator the dropout-mask cases compare against"""
    import numpy as np
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kat:
        got = kc.philox4x32_10_np(np.array([ctr], np.uint32), key)[0]
        assert tuple(int(v) for v in got) == exp


def test_dropout_mask(k):
    kc.case_dropout_mask(k, 1027, 0.1, 233, 0)
    kc.case_dropout_mask(k, 7 * 8 * 32, 0.1, (5 << 32) | 77, 12)


def test_adam_dev(k):
    kc.case_adam_dev(k, 1027)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape", [(3, 16, 64, 3, 3), (2, 9, 32, 2, 2), (3, 4, 160, 3, 2)])
def test_gru_chain(k, dtype, shape):
    kc.case_gru_chain(k, dtype, *shape)  # 48 rows: one full and one ragged row tile; 64/160 channels: one / two tiles per wave


@pytest.mark.parametrize("waves", ["8", "4"])
def test_gru_chain_full_width(k, monkeypatch, waves):
    """D = 256 (the reference's feature size): the eight-wave kernels (one column tile per wave; default) and the four-wave form"""
    monkeypatch.setenv("DPC_GRU_WAVES", waves)
    kc.case_gru_chain(k, BF16, 3, 12, 256, 2, 2)   # 36 rows: one full, one ragged row tile
    kc.case_gru_chain_philox(k, BF16, 2, 16, 256, 2, 1)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_gru_chain_philox(k, dtype):
    kc.case_gru_chain_philox(k, dtype, 3, 16, 64, 2, 2)


def test_gru_chain_reference_fixture(k, golden_dir):
    import os
    import numpy as np
    kc.case_gru_chain_golden(k, np.load(os.path.join(golden_dir, "ops.npz")))


@pytest.mark.parametrize("rd", [(24, 32), (96, 32), (200, 32), (264, 32), (136, 256)])
def test_score_fused(k, rd):
    kc.case_score_fused(k, *rd)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape", [(2, 16, 32, 1, 8, 8, (1, 1, 1), (1, 2, 2), (0, 0, 0)), (2, 16, 24, 5, 6, 6, (1, 1, 1), (2, 2, 2), (0, 0, 0)), (1, 16, 16, 2, 7, 9, (1, 3, 3), (1, 2, 2), (0, 1, 1))])
def test_conv_dgrad_inplace(k, dtype, shape):
    kc.case_conv_dgrad_inplace(k, dtype, *shape)


def test_stem_wgrad_fused(k):
    kc.case_stem_wgrad_fused(k, 2, 2, 128, 128)   # one 64-position segment per row
    kc.case_stem_wgrad_fused(k, 1, 1, 20, 200)    # ragged last segment, odd pooled width


@pytest.fixture
def x6(k):
    """f32 kernels in "bf16x6" arithmetic (include/dpc_hip.h: dpc_set_f32_matmul) for the duration of a test"""
    prev = k.lib.call("dpc_set_f32_matmul", 1)
    yield
    k.lib.call("dpc_set_f32_matmul", prev)


def test_f32_kernels_in_bf16x6_arithmetic(k, x6):
    """every f32 contraction kernel with its operands split three ways onto the bf16 matrix pipe: same cases, same f32 tolerances
    (1e-4 of max-abs for the convolutions, 1e-5 for the GEMMs against f64) -- the split keeps the products down to 2^-24"""
    kc.case_conv_fwd(k, F32, 2, 16, 64, 2, 9, 9, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    kc.case_conv_fwd(k, F32, 1, 32, 32, 3, 6, 6, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    kc.case_conv_dgrad(k, F32, 2, 16, 64, 5, 8, 8, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    kc.case_conv_dgrad(k, F32, 1, 128, 128, 1, 10, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    kc.case_conv_wgrad(k, F32, 2, 64, 72, 3, 8, 8, (3, 3, 3), (2, 2, 2), (1, 1, 1))      # transposing generic kernel
    kc.case_conv_wgrad(k, F32, 2, 64, 64, 1, 8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1))      # transpose-read kernel (position-major tiles)
    kc.case_conv_wgrad(k, F32, 5, 8, 24, 2, 6, 6, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    kc.case_gemm_nt(k, F32, 200, 100, 264)
    assert kc.case_gemm_nt_splitk(k, F32, 130, 64, 1024, pad=8) > 1
    with pytest.raises(L.DpcError):
        k.lib.call("dpc_set_f32_matmul", 2)
