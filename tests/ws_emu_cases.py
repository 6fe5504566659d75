"""Cases for the loader/compute wave-specialised implicit GEMM (dpc_amd/csrc/conv_igemm_ws.hip) on the host
SIMT simulator.  Run by tests/test_ws_emu.py in a child process: the kernel's row threshold and program
count are read from the environment once per process (DPC_IGEMM_WS_MINROWS / DPC_IGEMM_WS_GM), so that
small shapes reach it and one workgroup walks several tiles (ring of LDS stages across tile boundaries)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kcases as kc  # noqa: E402
from dpc_amd import _lib as L  # noqa: E402

BF16 = torch.bfloat16


def main():
    assert os.environ.get("DPC_IGEMM_WS_MINROWS") == "1"
    k = kc.K(L.load_emulator(), "cpu")
    # forward: 3x3 / 3x3x3 / strided 1x1, ragged last tile, Co = 128 and a ragged second column tile
    kc.case_conv_fwd(k, BF16, 3, 64, 128, 2, 10, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1))      # 960 rows: 4 tiles, 2 programs
    kc.case_conv_fwd(k, BF16, 2, 64, 136, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1))        # two column tiles
    kc.case_conv_fwd(k, BF16, 5, 128, 128, 2, 9, 7, (1, 1, 1), (1, 2, 2), (0, 0, 0))       # one tap, strided rows
    # temporally grouped tiles (3x3x3, power-of-two planes): padding taps skipped as a K sub-range
    kc.case_conv_fwd(k, BF16, 9, 64, 128, 2, 4, 4, (3, 3, 3), (1, 1, 1), (1, 1, 1))        # T = 2, 16 planes per tile
    kc.case_conv_dgrad(k, BF16, 3, 128, 64, 3, 4, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    # unit-stride input gradient (+ residual addend): the GEMM's columns are the conv's input channels
    kc.case_conv_dgrad(k, BF16, 2, 128, 64, 2, 9, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    kc.case_conv_dgrad(k, BF16, 1, 128, 128, 3, 6, 6, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    # plane variant (igemm_wsp_kernel): 16 x 16 planes staged as patches, K order (channel group, tap); 2 programs walk 2-3 tiles
    kc.case_conv_fwd(k, BF16, 3, 128, 128, 2, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1))      # layer2's shape: two channel groups
    kc.case_conv_fwd(k, BF16, 5, 64, 136, 1, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1))       # one group, odd tile count, ragged column tile
    kc.case_conv_dgrad(k, BF16, 2, 128, 256, 2, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1))    # four groups + residual addend
    kc.case_conv_dgrad_inplace(k, BF16, 1, 128, 128, 1, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1))  # a single tile
    # temporally grouped tiles whose 256 rows straddle clips (plane sizes that do not divide 256: the 224-pixel family)
    kc.case_conv_fwd(k, BF16, 7, 64, 128, 3, 7, 7, (3, 3, 3), (1, 1, 1), (1, 1, 1))         # 49-pixel planes, T = 3
    kc.case_conv_fwd(k, BF16, 3, 64, 128, 2, 14, 14, (3, 3, 3), (1, 1, 1), (1, 1, 1))       # 196-pixel planes, T = 2
    kc.case_conv_dgrad(k, BF16, 4, 128, 64, 3, 7, 7, (3, 3, 3), (1, 1, 1), (1, 1, 1))       # flipped taps + residual addend
    # 3x3x3 over 8 x 8 planes on the temporally grouped tiles of igemm_ws_kernel (layer3's shape)
    kc.case_conv_fwd(k, BF16, 9, 64, 128, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1))         # T = 3: border frames skip a tap; ragged last tile (9 clips)
    kc.case_conv_fwd(k, BF16, 4, 128, 136, 2, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1))        # T = 2, two channel groups, ragged column tile
    kc.case_conv_dgrad(k, BF16, 5, 128, 128, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1))      # flipped taps + residual addend
    # strided input-gradients by parity classes on the loader / compute kernel (igemm_ws_kernel<false, true>): 256-row tiles of one
    # class, K = the class's own taps; 2 programs walk the tiles of all classes (ring of stages across classes with 1..8 taps)
    PAR = "igemm_ws_kernel<false,true>"
    kc.case_conv_dgrad(k, BF16, 3, 128, 64, 3, 8, 8, (3, 3, 3), (2, 2, 2), (1, 1, 1), expect=PAR, with_add=False)      # 8 classes, temporal classes of 2 / 1 frames
    kc.case_conv_dgrad(k, BF16, 2, 128, 128, 2, 16, 36, (1, 3, 3), (1, 2, 2), (0, 1, 1), expect=PAR, with_add=False)   # 2D stride, two channel groups, 3 tiles per class (tail order)
    kc.case_conv_dgrad(k, BF16, 1, 136, 64, 1, 64, 144, (1, 3, 3), (1, 2, 2), (0, 1, 1), expect=PAR, with_add=False)   # 9 tiles per class: interleaved class order; ragged column tile
    kc.case_conv_dgrad(k, BF16, 2, 128, 64, 3, 7, 9, (3, 3, 3), (2, 2, 2), (1, 1, 1), expect=PAR, with_add=False)      # odd extents: unequal classes, class after class
    kc.case_conv_dgrad(k, BF16, 2, 64, 128, 1, 16, 24, (1, 3, 3), (1, 2, 2), (0, 1, 1), expect=PAR, with_add=False)     # 64 output columns on the 128-column tile
    # strided input-gradient over 16 x 16 gradient planes as a 2 x 2 unit-stride convolution onto 4 classes x 64 columns
    # (igemm_wsd_kernel: layer2.0.conv1 of the 128 x 128 configurations); one workgroup of each kind walks 4 / 5 planes
    kc.case_conv_dgrad(k, BF16, 2, 64, 128, 2, 32, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), expect="igemm_wsd_kernel", with_add=False)
    kc.case_conv_dgrad(k, BF16, 5, 64, 128, 1, 32, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), seed=11, expect="igemm_wsd_kernel", with_add=False)
    # the generic kernel's class order with several rounds per program (DPC_IGEMM_GM_CAP = 32 programs, 72 tiles): the class of a
    # block rotates with the round
    kc.case_conv_dgrad(k, BF16, 1, 32, 64, 1, 64, 144, (1, 3, 3), (1, 2, 2), (0, 1, 1), expect="igemm_kernel<T,TO,BN,3>", with_add=False)
    # role-specialised patch kernel (conv_halo_ws_kernel): DPC_HALO_WS_GM = 3 workgroups walk 24 / 8 tiles each
    kc.case_conv_fwd(k, BF16, 2, 64, 64, 2, 9, 33, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    kc.case_conv_fwd(k, BF16, 1, 64, 40, 1, 20, 12, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    kc.case_conv_dgrad(k, BF16, 2, 64, 64, 2, 9, 33, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    kc.case_stem(k, BF16, 2, 2, 16, 72)   # stem geometry (4x4 taps, 32-byte positions) on the role-specialised patch kernel
    # plain NT GEMMs
    kc.case_gemm_nt(k, BF16, 600, 136, 256)
    kc.case_gemm_nt(k, BF16, 257, 128, 64)
    print("ws cases ok")


if __name__ == "__main__":
    main()
