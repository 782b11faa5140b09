"""unet_amd: MI355X (gfx950) implementation of the U-Net layer operators behind
the Python module `neural_net`.

  unet_amd.capi   ctypes binding of the C ABI in include/unet_hip.h
  unet_amd.build  in-tree build (hipcc for the kernels, g++ for the module)
  unet_amd.dp     data-parallel training glue over torch.distributed (RCCL)
"""
__all__ = ["capi", "build"]
