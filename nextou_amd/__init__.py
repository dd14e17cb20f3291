"""nextou_amd — MI355X-native (gfx950) implementation of the NexToU forward/backward hot path.

Layout (only what the path needs):
  csrc/                  HIP kernels + the C-ABI of libnextou_hip.so (include/nextou_hip.h)
  graph_ops.py           host-side operators over the C-ABI (autograd glue)
  network_architecture/  mirror of the reference's nn.Module surface (NexToU, graph blocks)
  loss/                  BTI loss on the critical-voxel kernel + compound loss
  nnUNetTrainer/         nnU-Net v2 trainer plug-ins (names found by nnU-Net's class lookup)
  harness.py, ddp.py     standalone train step and RCCL data-parallel wrapper
"""
__version__ = "0.1.0"
