"""MI355X-native DPC-RNN training step (see README.md / DESIGN.md)."""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory (a ROCm runtime flag, read when HIP initialises): the ~600
# launches of a train step fetch their arguments from HBM instead of across PCIe.  Measured on MI355X (profiles/r04_probes.txt):
# hipGraph replay 25.95 -> 25.66 ms per step (+1.1 %), kernel-by-kernel launches 27.71 -> 27.49 ms.  Only a default: an explicit
# setting of the variable wins, and it has no effect once the process has touched the GPU (import dpc_amd first).
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
