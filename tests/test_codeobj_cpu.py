"""Static checks on the built gfx950 code objects (tools/codeobj_check.py): they guard two hand-counted synchronisation schemes
against a compiler upgrade or an innocent edit.  No GPU needed; needs the in-tree build (__graft_entry__.build())."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "audio8_amd", "csrc")


def _need(name):
    p = os.path.join(CSRC, name)
    if not os.path.exists(p):
        import __graft_entry__ as g
        g.build()
    assert os.path.exists(p), p
    return p


def test_stream_gemm_instantiations_the_library_picks_have_no_scratch():
    """csrc/gemm_stream.hip waits with hand-counted `s_waitcnt vmcnt(4)`; a scratch reload is a VMEM operation too, so a spilling
    instantiation drains its DMA ring every iteration (a silent performance cliff).  pick_bm offers tile heights 160 / 192 / 224
    (MI = 5, 6, 7): every such instantiation must be spill-free.  (MI = 8 is reachable only through a8_set_tuning key 7.)"""
    import codeobj_check as cc
    notes = cc.kernel_notes(_need("gemm_stream.o"))
    picked = {k: v for k, v in notes.items() if "gemm_stream_kernel" in k and any(f"ILi{mi}E" in k for mi in (5, 6, 7))}
    assert len(picked) == 12, sorted(notes)
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
