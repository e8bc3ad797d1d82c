"""Static checks on the built gfx950 code objects (tools/codeobj_check.py): they guard two hand-counted synchronisation schemes
against a compiler upgrade or an innocent edit.  No GPU needed; needs the in-tree build (__graft_entry__.build())."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "audio8_amd", "csrc")


LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def _need(name):
    """The object file of csrc/<name>; the tests SKIP on a box without the ROCm LLVM tools or hipcc (they are static checks of a build,
    not of the CPU-side logic)."""
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf", "llvm-objdump"):
        if not os.path.exists(os.path.join(LLVM_BIN, tool)):
            pytest.skip(f"{tool} not found under {LLVM_BIN}: no ROCm toolchain on this box")
    p = os.path.join(CSRC, name)
    if not os.path.exists(p):
        if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
            pytest.skip("no hipcc and no built objects")
        import __graft_entry__ as g
        g.build()
    if not os.path.exists(p):
        pytest.skip(f"{p} was not built")
    return p


def _demangled(names):
    """{mangled: demangled} through llvm-cxxfilt / c++filt (template arguments by name, not by their position-dependent mangling)."""
    import shutil
    import subprocess
    tool = os.path.join(LLVM_BIN, "llvm-cxxfilt")
    if not os.path.exists(tool):
        tool = shutil.which("c++filt")
    if tool is None:
        pytest.skip("no demangler (llvm-cxxfilt / c++filt) on this box")
    out = subprocess.run([tool] + list(names), check=True, capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def test_stream_gemm_instantiations_the_library_picks_have_no_scratch():
    """csrc/gemm_stream.hip waits with hand-counted `s_waitcnt vmcnt(4)`; a scratch reload is a VMEM operation too, so a spilling
    instantiation drains its DMA ring every iteration (a silent performance cliff).  pick_bm offers tile heights 160 / 192 / 224
    (MI = 5, 6, 7): every such instantiation must be spill-free.  (MI = 8 is reachable only through a8_set_tuning key 7.)"""
    import codeobj_check as cc
    notes = cc.kernel_notes(_need("gemm_stream.o"))
    import re
    dm = _demangled([k for k in notes if "gemm_stream_kernel" in k])
    # gemm_stream_kernel<MI, NA, ZIN>: tile heights 5 / 6 / 7 x ring depths 2 / 3 x with / without a second input
    picked = {k: notes[k] for k, n in dm.items() if (m := re.search(r"gemm_stream_kernel<(\d+), (\d+), (true|false)>", n)) and int(m.group(1)) in (5, 6, 7)}
    assert len(picked) == 3 * 2 * 2, sorted(dm.values())
    for k, v in picked.items():
        assert v.get("private_segment_fixed_size", 0) == 0 and v.get("vgpr_spill_count", 0) == 0, (k, v)
        assert v["vgpr_count"] <= 256, (k, v)


def test_gemm_kernels_meant_for_two_blocks_per_cu_fit_128_registers_without_scratch():
    """The 128 x 128 bf16 kernels rely on two co-resident 8-wave blocks per CU (4 waves per SIMD): <= 128 VGPRs, <= 80 KB of LDS, no scratch."""
    import codeobj_check as cc
    notes = cc.kernel_notes(_need("gemm.o"))
    a3 = {k: v for k, v in notes.items() if "gemm_a3" in k or "gemm_grouped_a3" in k}
    assert a3, sorted(notes)
    for k, v in a3.items():
        if "IfE" in k:                      # (f32 instantiations of the bf16-only ring are never launched)
            continue
        assert v["vgpr_count"] <= 128 and v.get("private_segment_fixed_size", 0) == 0, (k, v)
        assert v["group_segment_fixed_size"] <= 80 * 1024, (k, v)


def test_no_use_of_a_transposed_lds_read_before_its_wait():
    """StageTile<bf16>::ks_frag issues ds_read_b64_tr_b16 as `asm volatile` and the main loop waits with a hand-written
    s_waitcnt lgkmcnt(0): nothing between a read and that wait may touch the read's destination registers."""
    import codeobj_check as cc
    reads, bad = cc.tr_read_hazards(_need("gemm.o"))
    assert reads >= 64, reads                    # the K-strided instantiations are in the object at all
    assert not bad, bad[:5]


def test_dma_ring_attention_kernels_and_ctc_have_no_scratch():
    """The 32 x 32-MFMA attention kernels (attn_fwd32 and the round-6 backward pair) count their LDS-DMA ring with `s_waitcnt vmcnt(4 / 5)`
    by hand: a scratch reload in flight shifts the count.  Round 6 met two ways to get one silently -- a lambda whose by-reference capture
    is assigned between calls keeps its closure in scratch, `cond ? a[j] : b[j]` on two arrays becomes a dynamic index -- so the built
    kernels are checked: no private segment, no spills, and the register budgets their occupancy was planned for.  The CTC recurrence
    kernels (one or four waves per utterance, latency-bound) must not spill at any states-per-thread count either."""
    import codeobj_check as cc
    notes = cc.kernel_notes(_need("attention.o"))
    want = {"attn_fwd32_kernel": 168, "attn_bwd_q32_kernel": 168, "attn_bwd_kv32_kernel": 256}
    seen = {k: 0 for k in want}
    for k, v in notes.items():
        for name, cap in want.items():
            if name in k:
                seen[name] += 1
                assert v.get("private_segment_fixed_size", 0) == 0 and v.get("vgpr_spill_count", 0) == 0, (k, v)
                assert v["vgpr_count"] <= cap, (k, v)
    assert all(n >= 2 for n in seen.values()), seen
    ctc = {k: v for k, v in cc.kernel_notes(_need("ctc.o")).items() if "ctc_ab_kernel" in k}
    assert len(ctc) == 6, sorted(ctc)          # <states per thread, waves>: <1,1> <2,1> <1,4> <2,4> <3,4> <4,4>
    for k, v in ctc.items():
        assert v.get("private_segment_fixed_size", 0) == 0 and v.get("vgpr_spill_count", 0) == 0, (k, v)
