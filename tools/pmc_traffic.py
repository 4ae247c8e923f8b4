#!/usr/bin/env python3
"""HBM traffic per launch of the conv GEMM kernels from two rocprofv3 --pmc passes of bench.py
(FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots), corrected as that
guide prescribes for gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide streaming read -> x2;
WRITE_SIZE is exact for 16-B stores; both are reported in KiB by rocprofv3.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <config> > profiles/rNN_<config>_pmc_traffic.json
"""
import csv
import json
import sys


def per_kernel(path, counter):
    agg = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            key = ("igemm3" if "igemm3" in name else "igemm2" if "igemm2_kernel" in name
                   else "wgrad3" if "wgrad3_kernel" in name else "wgrad2" if "wgrad2_kernel" in name
                   else "enet_conv" if "enet_conv_kernel" in name else "enet_reduce" if "enet_reduce" in name
                   else "enet_wgrad" if "enet_wgrad_kernel" in name else "enet_finalize" if "finalize_kernel" in name and "enet" in name
                   else "enet_bn_bwd_apply" if "enet_bn_bwd_apply" in name else "enet_tail" if "enet_tail" in name
                   else "enet_wgrad_reduce" if "enet_wgrad_reduce" in name
                   # round 5: the launches around the conv kernels too, so that bytes moved OUT of the conv families (un-pooling, folds) stay visible
                   else "zz_folds" if ("wgrad_reduce" in name or "splitk_epilogue" in name or "slab_rows_fold" in name or "split_dw_db" in name)
                   else "zz_pool_bilinear" if ("maxpool" in name or "bilinear" in name) else "zz_adam" if "adam_kernel" in name
                   else "zz_other")
            a = agg.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write, config = sys.argv[1], sys.argv[2], sys.argv[3]
    launches_total = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # profiled steps incl. set-up / warm-up (per-step totals)
    fa, wa = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    out = {"unit": "bytes per launch (HBM-side, L2 fabric counters)", "config": config,
           "correction": "FETCH_SIZE x 1024 (KiB) x 2 (gfx950 wide-read undercount); WRITE_SIZE x 1024"}
    for k in sorted(fa):
        n, v = fa[k]
        nw, vw = wa.get(k, [0, 0.0])
        out[k] = {"launches": n, "fetch_bytes_per_launch": v * 1024 * 2 / max(n, 1),
                  "write_bytes_per_launch": vw * 1024 / max(nw, 1)}
        out[k]["bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
    if launches_total:
        conv = sum(v["bytes_per_launch"] * v["launches"] for k, v in out.items() if isinstance(v, dict) and not k.startswith("zz_"))
        rest = sum(v["bytes_per_launch"] * v["launches"] for k, v in out.items() if isinstance(v, dict) and k.startswith("zz_"))
        out["all_listed_kernels"] = {"bytes_per_step": conv / launches_total, "steps_profiled": launches_total,
                                     "what": "the conv / weight-gradient kernel families (the figure of rounds 1-4)"}
        out["whole_step"] = {"bytes_per_step": (conv + rest) / launches_total, "what": "every launch of the step (zz_* = the launches around the conv kernels)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
