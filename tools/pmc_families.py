#!/usr/bin/env python3
"""Per-kernel-family HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; own runs, kernel-trace only), per the
HBM section of MI355X_MICROARCH.md: KiB units; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced stream's bytes (doubled here),
WRITE_SIZE is exact.  usage: pmc_families.py FETCH_DIR WRITE_DIR STEPS OUT_JSON [CONV_FETCH_DIR CONV_WRITE_DIR CONV_PASSES]"""
import collections
import csv
import glob
import json
import sys

FAMILIES = [("gemm", lambda k: ("gemm_" in k and "kernel" in k) and "<float" not in k and "IfL" not in k),
            ("conv0_apply", lambda k: "conv0_apply" in k), ("conv0_bwd", lambda k: "conv0_bwd_m" in k), ("conv0_moments", lambda k: "conv0_moments" in k),
            ("ln_fwd", lambda k: "ln_fwd" in k), ("ln_bwd", lambda k: "ln_bwd" in k),
            ("attn_fwd", lambda k: "attn_fwd" in k), ("attn_bwd_q", lambda k: "attn_bwd_q" in k), ("attn_bwd_kv", lambda k: "attn_bwd_kv" in k),
            ("adamw", lambda k: "adamw" in k), ("sumsq", lambda k: "sumsq" in k), ("wprep", lambda k: "wprep" in k),
            ("posconv", lambda k: "posconv" in k), ("gelu_bwd_bias", lambda k: "gelu_bwd_bias" in k)]


def rows(d, counter):
    out = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        out += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    return out


def main():
    fd, wd, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fr, wr = rows(fd, "FETCH_SIZE"), rows(wd, "WRITE_SIZE")
    fam = collections.OrderedDict()
    for name, pred in FAMILIES:
        f = [float(r["Counter_Value"]) * 1024 * 2 for r in fr if pred(r["Kernel_Name"])]
        w = [float(r["Counter_Value"]) * 1024 for r in wr if pred(r["Kernel_Name"])]
        if not f:
            continue
        fam[name] = {"launches_per_step": round(len(f) / steps, 2), "fetch_bytes_per_launch": round(sum(f) / len(f)),
                     "write_bytes_per_launch": round(sum(w) / max(1, len(w))), "hbm_bytes_per_step": round((sum(f) + sum(w)) / steps)}
    g = fam["gemm"]
    res = {"kernel": "every bf16 a8_gemm / a8_gemm_grouped kernel (gemm_kernel, gemm_grouped_kernel and their asymmetric-ring forms)", "steps_profiled": steps,
           "fetch_bytes_per_launch": g["fetch_bytes_per_launch"], "write_bytes_per_launch": g["write_bytes_per_launch"],
           "hbm_bytes_per_launch": g["fetch_bytes_per_launch"] + g["write_bytes_per_launch"], "hbm_bytes_per_step": g["hbm_bytes_per_step"],
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes); KiB units; FETCH_SIZE x2 (gfx950)",
           "families": fam}
    if len(sys.argv) > 7:
        cf, cw, n = rows(sys.argv[5], "FETCH_SIZE"), rows(sys.argv[6], "WRITE_SIZE"), int(sys.argv[7])
        res["conv_extractor_fwd_hbm_bytes"] = round((sum(float(r["Counter_Value"]) for r in cf) * 2048 + sum(float(r["Counter_Value"]) for r in cw) * 1024) / n)
        res["conv_extractor_fwd_note"] = "all kernels of tools/conv_only.py (conv stack forward only, %d passes), same counters and corrections" % n
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
