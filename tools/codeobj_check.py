"""Static checks on the built gfx950 code objects (no GPU needed): per-kernel register / scratch notes and a scan for uses of an
inline-asm ds_read_b64_tr_b16 destination ahead of the s_waitcnt lgkmcnt(0) that retires it.

Why: (1) csrc/gemm_stream.hip counts its DMA waits by hand (s_waitcnt vmcnt(4)); scratch reloads count in vmcnt too, so a spilling
instantiation silently drains its ring every iteration -- the instantiations the library can pick must have no scratch.  (2) csrc/gemm.hip
issues the transposed LDS reads of the K-strided layouts as `asm volatile` and waits for them by hand; the compiler believes the results
are ready at once, so any instruction it schedules between the reads and the wait that touches their destinations reads stale registers.
Used by tests/test_codeobj_cpu.py; `python tools/codeobj_check.py` prints the tables."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio8_amd", "csrc")


def device_code_object(obj_path, workdir):
    """gfx950 code object embedded in a hipcc host object (its .hip_fatbin section)."""
    fb = os.path.join(workdir, os.path.basename(obj_path) + ".fatbin")
    co = os.path.join(workdir, os.path.basename(obj_path) + ".co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj_path, fb], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    return co


def kernel_notes(obj_path):
    """{kernel name: {vgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size}}"""
    with tempfile.TemporaryDirectory() as td:
        co = device_code_object(obj_path, td)
        txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "agpr_count":          # first key of a kernel's record in the metadata map
            cur = {}
        if cur is None:
            continue
        if k == "name":
            out[v] = cur
        elif k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count"):
            cur[k] = int(v)
    return out


_REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def _regs(text):
    s = set()
    for m in _REG.finditer(text):
        if m.group(3) is not None:
            s.add(int(m.group(3)))
        else:
            s.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return s


def tr_read_hazards(obj_path):
    """Instructions that touch the destination of a ds_read_b64_tr_b16 before the next `s_waitcnt ... lgkmcnt(0)`.
    Returns (number of transposed reads seen, [(kernel, address, instruction)])."""
    with tempfile.TemporaryDirectory() as td:
        co = device_code_object(obj_path, td)
        txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    pending, kernel, reads, bad = set(), None, 0, []
    for line in txt.splitlines():
        lab = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if lab:
            kernel, pending = lab.group(1), set()
            continue
        m = re.match(r"^\s*(\S+)\s*(.*?)\s*(?://\s*([0-9A-Fa-f]+):.*)?$", line)
        if not m or not m.group(1) or m.group(1).endswith(":"):
            continue
        op, args, addr = m.group(1), m.group(2), m.group(3)
        if op == "ds_read_b64_tr_b16":
            reads += 1
            dst = args.split(",")[0]
            pending |= _regs(dst)
            continue
        if op == "s_waitcnt":
            if "lgkmcnt(0)" in args:
                pending = set()
            continue
        if op.startswith("s_") and not pending:
            continue
        if pending and (_regs(args) & pending):
            bad.append((kernel, addr, f"{op} {args}"))
    return reads, bad


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["gemm_stream.o", "gemm.o", "attention.o"]):
        p = os.path.join(CSRC, name)
        print(f"== {name}")
        for k, v in kernel_notes(p).items():
            print(f"  {k[:90]:90s} vgpr {v.get('vgpr_count', -1):3d} spill {v.get('vgpr_spill_count', 0):2d} scratch {v.get('private_segment_fixed_size', 0):3d} "
                  f"lds {v.get('group_segment_fixed_size', 0)}")
    n, bad = tr_read_hazards(os.path.join(CSRC, "gemm.o"))
    print(f"gemm.o: {n} ds_read_b64_tr_b16, {len(bad)} destination uses ahead of their lgkmcnt(0)")
    for b in bad[:20]:
        print("  ", b)
