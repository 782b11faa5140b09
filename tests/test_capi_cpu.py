"""CPU-only checks of the drop-in boundary: the shared library loads and
exports every symbol include/unet_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from unet_amd import capi


def test_header_parses():
    protos = capi.PROTOTYPES
    assert len(protos) >= 45
    for must in ["unet_convolve_f32", "unet_convolve_backward_f32", "unet_batch_norm_train_f32",
                 "unet_max_pool_indices_f32", "unet_upsample_backward_f32", "unet_concat_f32",
                 "unet_dice_score_f32", "unet_update_weights_adam_f32", "unet_last_error"]:
        assert must in protos, must
    # the conv entry point keeps the argument list of cudaConvolve (cuda_lib.h:44-58) + stream
    names = [n for n, _ in protos["unet_convolve_f32"][1]]
    assert names == ["weights", "inputs", "activations", "activation_array_length", "batch_size",
                     "kernel_size", "channels_in", "channels_out", "H_out", "W_out", "H_in", "W_in",
                     "padding", "stride", "stream"]


def test_library_exports_every_declared_symbol():
    assert os.path.exists(capi.LIB_PATH), "build with `python -m unet_amd.build`"
    lib = ctypes.CDLL(capi.LIB_PATH)
    missing = [n for n in capi.PROTOTYPES if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert capi.unet_version() >= 100


def test_every_exported_unet_symbol_is_declared():
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (unet_\w+)", out))
    internal = {"unet_set_error", "unet_workspace", "unet_workspace_release_all"}
    undeclared = sorted(n for n in exported if n not in capi.PROTOTYPES and n not in internal
                        and not n.startswith("_Z"))
    assert not undeclared, f"exported but not in include/unet_hip.h: {undeclared}"


def test_argument_validation_needs_no_gpu():
    # status codes and messages come back through the C ABI
    import pytest
    with pytest.raises(RuntimeError, match="NULL"):
        capi.unet_convolve_f32(None, None, None, 0, 1, 3, 1, 1, 4, 4, 4, 4, 1, 1, None)
    assert "NULL" in capi.last_error()
