"""Fabric-side traffic of the Enet kernel families per step from two rocprofv3 counter passes restricted to them:
    rocprofv3 --pmc TCC_EA0_RDREQ_sum --kernel-trace --kernel-include-regex "$RX" ... -- python3 bench.py --config cfg4 ...
    rocprofv3 --pmc WRITE_SIZE        --kernel-trace --kernel-include-regex "$RX" ... -- python3 bench.py --config cfg4 ...
(the derived FETCH_SIZE metric crashes the tool on this workload; TCC_EA0_RDREQ_sum is its base counter: requests x 64 B, a
lower bound -- gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md).
    python tools/pmc_traffic_enet.py <read counter csv> <write counter csv> [conv launches per step = 966] > out.json"""
import collections
import csv
import json
import re
import sys


def load(path, cname):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != cname:
                continue
            # round 3: every Enet kernel is enet_one<F> / enet_grp<F> of a body functor F (MconvK, ReduceVecK, BnFinK, ...)
            m = re.search(r"(Mconv|Mwgrad|ReduceVec|Reduce|BnBwdFin|BnFin|BnApplyVec|BnApply|WgradRed|Wgrad|Conv|TailFwd|TailBwd|SumFin)K", r["Kernel_Name"])
            if m is None:
                m = re.search(r"enet_[a-z_]+", r["Kernel_Name"])
            k = m.group(0) if m else r["Kernel_Name"][:30]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    rd, wr = load(sys.argv[1], "TCC_EA0_RDREQ_sum"), load(sys.argv[2], "WRITE_SIZE")
    per_step = float(sys.argv[3]) if len(sys.argv) > 3 else 966.0
    conv = "MconvK" if "MconvK" in rd else "enet_mconv_kernel"
    steps = rd[conv][0] / per_step
    out, tr, tw = {}, 0.0, 0.0
    for k in sorted(rd, key=lambda k: -rd[k][1]):
        r, w = rd[k][1] * 64 / steps, wr.get(k, [0, 0])[1] * 1024 / steps
        tr, tw = tr + r, tw + w
        out[k] = {"launches_per_step": rd[k][0] / steps, "read_requests_x64B_MB_per_step": r / 1e6, "write_MB_per_step": w / 1e6}
    json.dump({"unit": "MB per step, HBM-side (L2 fabric counters), Enet kernel families only", "steps": steps,
               "read_bytes_rule": "TCC_EA0_RDREQ_sum x 64 B (lower bound; wide streaming reads are up to 2x this on gfx950)",
               "kernels": out, "total_read_MB_lower_bound": tr / 1e6, "total_write_MB": tw / 1e6}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
