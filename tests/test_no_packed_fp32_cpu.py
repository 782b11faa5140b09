"""The built device code holds no packed fp32 arithmetic (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).

Those instructions returned wrong values on MI355X in waves that shared a SIMD with waves of this library's
MFMA kernels (unet_amd/csrc/bf16_util.h); the build switches the feature off for device code
(unet_amd/build.py), and this test disassembles every gfx950 code object of the shipped library to keep it so.
Needs the ROCm LLVM tools only, no GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import scan_code_objects as sco  # noqa: E402

pytestmark = pytest.mark.skipif(not (os.path.exists(sco.LIB) and os.path.exists(os.path.join(sco.LLVM, "llvm-objdump"))),
                                reason="needs the built library and the ROCm LLVM tools")


def test_no_packed_fp32_arithmetic_in_any_kernel():
    hits, functions, objects = sco.count_per_kernel(sco.PACKED_FP32)
    assert objects >= 7 and functions >= 150, (objects, functions)   # the scan did see the library's kernels
    assert not hits, f"packed fp32 arithmetic in: {sorted(hits.items(), key=lambda kv: -kv[1])[:10]}"


def test_the_scan_sees_instructions():
    # the same scan finds what is there: matrix-core instructions in the convolution kernels, and no calls
    # (everything inlines; a call would mean the build's feature switch broke inlining)
    mfma, _, _ = sco.count_per_kernel(r"\bv_mfma_f32_16x16x32_bf16\b")
    assert any("conv3x3_pipe_kernel" in k for k in mfma)
    calls, _, _ = sco.count_per_kernel(r"\bs_swappc_b64\b")
    assert not calls, list(calls)[:5]


def test_the_pipelined_convolution_keeps_everything_in_registers():
    # a spilled register in conv3x3_pipe_kernel is reloaded through vector memory, whose counter the kernel
    # uses for its direct-to-LDS loads: such a reload waits for every operand in flight (measured: 10-80 %
    # slower), and an out-of-line lambda would put the accumulators into scratch altogether
    scratch, _, _ = sco.count_per_kernel(r"\bscratch_(load|store)_")
    spilled = {k: v for k, v in scratch.items() if "conv3x3_pipe_kernel" in k or "wgrad_dma_kernel" in k}
    assert not spilled, spilled
    # the transposed-convolution matrix product keeps 128 accumulators, two sets of weight fragments and a chunk of
    # staged pixels in 256 registers: a deeper fragment ring has put its fused-reduction instance into spills before
    spilled = {k: v for k, v in scratch.items() if "tconv_gemm_kernel" in k}
    assert not spilled, spilled
    # and every instance has its 144 MFMAs per stage body (first, middle, last, single stage), not fewer
    mfma, _, _ = sco.count_per_kernel(r"\bv_mfma_f32_16x16x32_bf16\b")
    pipe32 = {k: v for k, v in mfma.items() if "conv3x3_pipe_kernel" in k and k.endswith("Li32EEEvNS_4ArgsE")}
    assert pipe32 and all(v == 4 * 144 for v in pipe32.values()), pipe32
