#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per the HBM section
of MI355X_MICROARCH.md: counters are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
coalesced streaming read, so it is doubled; WRITE_SIZE is taken as is.  Output: profiles/<tag>_gemm_traffic.json."""
import csv
import glob
import json
import sys


def per_kernel(d, counter, match):
    rows = [r for r in csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])) if r["Counter_Name"] == counter]
    # every gemm launch except the explicit f32 instantiations (the trace mixes mangled and demangled names)
    sel = [r for r in rows if (match in r["Kernel_Name"] or "gemm_grouped_kernel" in r["Kernel_Name"]) and "<float" not in r["Kernel_Name"]
           and "gemm_kernelIfL" not in r["Kernel_Name"] and "gemm_grouped_kernelIfL" not in r["Kernel_Name"]]
    return sum(float(r["Counter_Value"]) for r in sel), len(sel)


def main():
    fetch_dir, write_dir, out, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    f, nf = per_kernel(fetch_dir, "FETCH_SIZE", "gemm_kernel")
    w, nw = per_kernel(write_dir, "WRITE_SIZE", "gemm_kernel")
    fetch_bytes = f * 1024 * 2           # gfx950 correction
    write_bytes = w * 1024
    res = {"kernel": "gemm_kernel<bf16> + gemm_grouped_kernel<bf16>", "launches_profiled": nf, "steps_profiled": steps,
           "fetch_bytes_per_launch": fetch_bytes / max(1, nf), "write_bytes_per_launch": write_bytes / max(1, nw),
           "hbm_bytes_per_launch": fetch_bytes / max(1, nf) + write_bytes / max(1, nw),
           "hbm_bytes_per_step": (fetch_bytes + write_bytes) / steps,
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes); KiB units; FETCH_SIZE x2 (gfx950)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
