"""ISA audit of the ping-pong GEMM kernels (CPU: hipcc cross-compiles gfx950 without a GPU).

The kernel's speed rests on two properties of the generated code that no numerical test sees:
no scratch traffic in the K loop (a spilled loop invariant comes back through a scratch load whose vmcnt(0)
drains the LDS-DMA stream) and no compiler-inserted vmcnt wait (a compiler-visible global access in the loop
makes hipcc put vmcnt(0) in front of fragment reads / MFMAs).  Both showed up during development as silent
10 - 30 % losses that depended on unrelated code, so the library build is checked for them here."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def gemm_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa")
    # the flags of mcm_amd/csrc/Makefile (no -DMCM_HARNESS: the shipped code, not the harness build)
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "mcm_amd", "csrc"),
           "-c", os.path.join(ROOT, "mcm_amd", "csrc", "gemm.hip"), "-o", str(out / "gemm.o"), "-save-temps=obj"]
    subprocess.run(cmd, check=True, cwd=str(out), capture_output=True, timeout=600)
    asm = [f for f in os.listdir(out) if f.endswith("gfx950.s")]
    assert asm, os.listdir(out)
    return open(out / asm[0]).read()


@pytest.fixture(scope="module")
def gemm_isa_harness(tmp_path_factory):
    """The shipped flags plus -DMCM_LN_FOLD -DMCM_LN_TAIL: the build the LayerNorm-fold / LayerNorm-tail arms were timed in (in the -DMCM_HARNESS build the
    ablation branches around every epilogue store cost the fp16 consumer form four spilled registers)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa_h")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DMCM_LN_FOLD", "-DMCM_LN_TAIL", "-I",
           os.path.join(ROOT, "mcm_amd", "csrc"), "-c", os.path.join(ROOT, "mcm_amd", "csrc", "gemm.hip"), "-o",
           str(out / "gemm.o"), "-save-temps=obj"]
    subprocess.run(cmd, check=True, cwd=str(out), capture_output=True, timeout=600)
    asm = [f for f in os.listdir(out) if f.endswith("gfx950.s")]
    assert asm, os.listdir(out)
    return open(out / asm[0]).read()


def _kernel(isa, prec, epi, fold=0, arms=None):
    """The shipped gemm_pp_kernel<PREC, EPI> (gemm.hip) or — with `fold` / `arms` — the flagged text of gemm_arms.hpp:
    arms::gemm_pp_kernel<PREC, EPI, FOLD = fold, LNT = false, LNC = false>."""
    if arms is None:
        arms = bool(fold)
    if arms:
        pat = r"^(_ZN\S*4arms14gemm_pp_kernelILi%dELi%dELb%dELb0ELb0EEEv8GemmArgs):\s.*?^\.Lfunc_end" % (prec, epi, fold)
    else:
        pat = r"^(_ZN\S*_114gemm_pp_kernelILi%dELi%dEEEv8GemmArgs):\s.*?^\.Lfunc_end" % (prec, epi)
    m = re.search(pat, isa, re.S | re.M)
    assert m, "gemm_pp_kernel<%d,%d,fold=%d,arms=%s> not found" % (prec, epi, fold, arms)
    return m.group(0)


@pytest.mark.parametrize("prec", [0, 2], ids=["bf16", "fp16"])
@pytest.mark.parametrize("epi", [0, 1], ids=["store", "gelu"])
@pytest.mark.parametrize("fold", [0, 1], ids=["plain", "ln-fold-consumer"])
def test_pingpong_16bit_kernels_have_no_scratch_and_only_the_two_hand_written_waits(gemm_isa, gemm_isa_harness, prec, epi, fold):
    body = _kernel(gemm_isa_harness if fold else gemm_isa, prec, epi, fold)
    assert "scratch_" not in body
    assert len(re.findall(r"s_waitcnt vmcnt", body)) == 2  # prologue + end of the compute phase
    assert len(re.findall(r"v_mfma_f32_16x16x32", body)) == 64  # one compute phase, no duplicated loop bodies


@pytest.mark.parametrize("prec", [0, 2], ids=["bf16", "fp16"])
def test_pingpong_residual_kernel_keeps_the_k_loop_clean_and_does_not_spill(gemm_isa, prec):
    body = _kernel(gemm_isa, prec, 2)
    lines = body.splitlines()
    mfma = [i for i, l in enumerate(lines) if "v_mfma_f32_16x16x32" in l]
    dma = [i for i, l in enumerate(lines) if "global_load_lds_dwordx4" in l]
    assert len(mfma) == 64
    # the hot loop = from the last block of LDS-DMA issues before the MFMAs to the last MFMA: no scratch there,
    # and no vmcnt wait between the DMA issues and the MFMAs
    start = max(i for i in dma if i < mfma[0]) - 200
    hot = lines[start:mfma[-1] + 1]
    assert not any("scratch_" in l for l in hot)
    between = lines[max(i for i in dma if i < mfma[0]):mfma[0]]
    assert not any("s_waitcnt vmcnt" in l for l in between)
    # round 3: the bias is loaded inside the epilogue (after the LDS bounce, 4 registers) instead of being carried
    # across the last compute phase (16): the four spilled registers of rounds 1 - 2 and their reload wait are gone
    assert sum("scratch_" in l for l in lines) == 0
    assert len(re.findall(r"s_waitcnt vmcnt\(0\)", body)) == 2


@pytest.mark.parametrize("prec", [0, 2], ids=["bf16", "fp16"])
def test_pingpong_ln_fold_producer_has_no_scratch_and_only_hand_counted_waits(gemm_isa_harness, prec):
    """The residual kernel with the LayerNorm-fold epilogue (z, row moments): nothing of its epilogue is live across
    the K loop (gamma and the bias are loaded inside it), so unlike the plain residual kernel it does not spill at all;
    every vmcnt wait is one of the hand-written counts (two copies of the epilogue: in the loop and after it)."""
    body = _kernel(gemm_isa_harness, prec, 2, 1)
    assert "scratch_" not in body
    waits = sorted(int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body))
    assert waits == sorted([0, 0] + 2 * [4, 12, 12, 12, 13, 12, 12, 8]), waits
    assert len(re.findall(r"v_mfma_f32_16x16x32", body)) == 64
    assert not re.search(r"v_fma_f32|v_fmac_f32|v_pk_fma_f32", body)  # the moments are explicitly rounded operations


@pytest.mark.parametrize("prec", [0, 2], ids=["bf16", "fp16"])
def test_pingpong_residual_kernel_with_the_layernorm_tail_keeps_the_k_loop_clean(gemm_isa_harness, prec):
    """gemm_pp_kernel<PREC, EPI_RESID, FOLD = false, LNT = true> (LayerNorm in the tail): the publication of a finished tile adds a
    uniform branch and one asm atomic behind the wait that ends a compute phase, the tail itself sits behind the loop —
    the K loop must look exactly like the plain residual kernel's: no scratch, 64 MFMAs, no wait between the LDS-DMA
    issues and the MFMAs, and up to the last MFMA only the hand-written waits."""
    m = re.search(r"^(_ZN\S*4arms14gemm_pp_kernelILi%dELi2ELb0ELb1ELb0EEEv8GemmArgs):\s.*?^\.Lfunc_end" % prec, gemm_isa_harness,
                  re.S | re.M)
    assert m, "LNT kernel not found"
    lines = m.group(0).splitlines()
    assert not any("scratch_" in l for l in lines)
    mfma = [i for i, l in enumerate(lines) if "v_mfma_f32_16x16x32" in l]
    dma = [i for i, l in enumerate(lines) if "global_load_lds_dwordx4" in l]
    assert len(mfma) == 64
    between = lines[max(i for i in dma if i < mfma[0]):mfma[0]]
    assert not any("s_waitcnt vmcnt" in l for l in between)
    plain = _kernel(gemm_isa_harness, prec, 2).splitlines()
    plain_mfma = [i for i, l in enumerate(plain) if "v_mfma_f32_16x16x32" in l]
    waits = lambda ls, end: [l.strip() for l in ls[:end] if "s_waitcnt vmcnt" in l]  # noqa: E731
    assert waits(lines, mfma[-1]) == waits(plain, plain_mfma[-1])


@pytest.mark.parametrize("prec", [0, 2], ids=["bf16", "fp16"])
def test_pingpong_residual_kernel_with_the_cluster_layernorm_keeps_the_k_loop_clean(gemm_isa_harness, prec):
    """gemm_pp_kernel<PREC, EPI_RESID, false, false, LNC = true> (round 6: the LayerNorm written by the residual epilogue from the
    accumulator registers).  Its epilogue holds the 128 accumulators live from the residual add to the LayerNorm store, which
    first cost the K loop its fragment offset (spilled, reloaded in the compute phase behind a vmcnt(0) that drains the LDS-DMA
    stream) — cured by fetching the epilogue's kernel arguments inside it and by deriving its late addresses from a fresh
    lane-id copy.  Held here: no scratch access from the LDS-DMA issues of a K-step to its last MFMA, no wait between the DMA
    issues and the MFMAs, 64 MFMAs, and whatever is still spilled (a lane id, the saturation watch) is touched on the
    tile-boundary path only: at most 12 scratch instructions in the whole kernel."""
    m = re.search(r"^(_ZN\S*4arms14gemm_pp_kernelILi%dELi2ELb0ELb0ELb1EEEv8GemmArgs):\s.*?^\.Lfunc_end" % prec, gemm_isa_harness,
                  re.S | re.M)
    assert m, "LNC kernel not found"
    lines = m.group(0).splitlines()
    mfma = [i for i, l in enumerate(lines) if "v_mfma_f32_16x16x32" in l]
    dma = [i for i, l in enumerate(lines) if "global_load_lds_dwordx4" in l]
    assert len(mfma) == 64
    start = max(i for i in dma if i < mfma[0])
    hot = lines[start - 150:mfma[-1] + 1]
    assert not any("scratch_" in l for l in hot), [l for l in hot if "scratch_" in l]
    assert not any("s_waitcnt vmcnt" in l for l in lines[start:mfma[0]])
    assert sum("scratch_" in l for l in lines) <= 12


@pytest.mark.parametrize("prec", [0, 1, 2], ids=["bf16", "fp32", "fp16"])
def test_pixel_gathering_patch_gemm_does_not_spill(gemm_isa, prec):
    """gemm_p256_kernel<PREC, EPI_PATCH, false, PXF = true> (the im2col-free patch embedding): 32 registers of pixel loads
    are in flight across a K-step of MFMAs next to 128 accumulators.  As a run-time branch of the plain kernel it spilled
    13 registers; as its own instantiation (no X row pointers, no bias registers) it must not touch scratch, must keep the
    W operand on the LDS-DMA path and fetch pixels with plain 16-byte loads."""
    m = re.search(r"^(_ZN\S*_116gemm_p256_kernelILi%dELi3ELb0ELb1EEEv8GemmArgs):\s.*?^\.Lfunc_end" % prec, gemm_isa, re.S | re.M)
    assert m, "pixel-gathering patch GEMM not found"
    body = m.group(0)
    assert "scratch_" not in body
    assert "global_load_lds_dwordx4" in body and len(re.findall(r"global_load_dwordx4 v", body)) >= 4
    assert len(re.findall(r"v_mfma_f32_16x16x(32|4)", body)) in (64, 256)


@pytest.fixture(scope="module")
def preprocess_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa_p")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "mcm_amd", "csrc"),
           "-c", os.path.join(ROOT, "mcm_amd", "csrc", "preprocess.hip"), "-o", str(out / "preprocess.o"), "-save-temps=obj"]
    subprocess.run(cmd, check=True, cwd=str(out), capture_output=True, timeout=600)
    asm = [f for f in os.listdir(out) if f.endswith("gfx950.s")]
    assert asm, os.listdir(out)
    return open(out / asm[0]).read()


def test_resize_kernel_keeps_lds_address_space_and_avoids_the_packed_shift(preprocess_isa):
    """Two properties of mcm_resize_crop_u8's LDS form that only the generated code shows (both cost a wrong image on the
    MI355X before the bit-exact tests caught them, preprocess.hip):
    * its LDS traffic is ds_* — no flat store (an LDS pointer re-aligned through an integer cast turns the accesses into
      flat ones, and hipcc then merges the horizontal pass's byte stores into 16-bit flat stores at odd addresses);
    * no v_ashr_pk_u8_i32: hipcc fuses two neighbouring (>> 22, clamp to 0..255) into it and ORs the result into a packed
      word as if the destination's upper half were zero, which it is not on this device."""
    m = re.search(r"^(_ZN\S*18resize_crop_kernel\S*):\s.*?^\.Lfunc_end", preprocess_isa, re.S | re.M)
    assert m, "resize_crop_kernel not found"
    body = m.group(0)
    assert "v_ashr_pk" not in preprocess_isa
    assert "flat_store" not in body and "scratch_" not in body
    assert body.count("ds_write_b8") >= 3 and "v_alignbyte_b32" in body and "v_mul_u32_u24" in body
    # nothing in the LDS form multiplies on the quarter-rate 32-bit multiplier: the only v_mul_lo_u32 left are the fused
    # form's taps and index arithmetic
    assert body.count("v_mul_u32_u24") + body.count("v_mad_u32_u24") > 100


def test_jpeg_kernels_have_no_scratch_and_no_packed_shift(tmp_path_factory):
    """jpeg.hip: the IDCT kernel keeps its 64 values in registers (a dynamic index into the by-value image record once sent
    it to scratch) and neither kernel may contain v_ashr_pk_u8_i32 (the (shift, clamp) pair hipcc fuses into it is exactly
    what both kernels end with; on the MI355X its upper half is not zero, preprocess.hip)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa_j")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "mcm_amd", "csrc"),
           "-c", os.path.join(ROOT, "mcm_amd", "csrc", "jpeg.hip"), "-o", str(out / "jpeg.o"), "-save-temps=obj"]
    subprocess.run(cmd, check=True, cwd=str(out), capture_output=True, timeout=600)
    isa = open(out / [f for f in os.listdir(out) if f.endswith("gfx950.s")][0]).read()
    assert "v_ashr_pk" not in isa and "scratch_" not in isa
    assert isa.count(".private_segment_fixed_size: 0") == 2   # both kernels
