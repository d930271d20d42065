#!/usr/bin/env python3
"""Fold rocprofv3 --pmc CSVs (one directory per pass) into per-kernel averages per dispatch.
FETCH_SIZE is doubled per MI355X_MICROARCH.md "HBM" (gfx950 tallies 128-B requests at 64 B for
wide coalesced reads); both raw and corrected values are kept."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    import re
    if "k_atrous_chain" in name:
        # k_atrous_chain<L, FINAL, EXACT>
        m = re.search(r"k_atrous_chain<\d+, (true|false)", name)
        return "k_atrous_chain_final" if (m and m.group(1) == "true") else "k_atrous_chain"
    if "k_pathtrace_binned" in name:
        return "k_pathtrace_binned"
    if "k_pathtrace_queue" in name:
        return "k_pathtrace_queue"
    if "k_atrous" in name:
        # k_atrous_comb_sh<CW, FINAL, EXACT> / k_atrous<FINAL, EXACT>
        m = re.search(r"k_atrous(?:_comb_sh<\d+, |<)(true|false)", name)
        return "k_atrous_final" if (m and m.group(1) == "true") else "k_atrous"
    if "k_gbuffer_pathtrace" in name:   # K0 + K1 + K2 in one launch (round 4)
        return "k_gbuffer_pathtrace"
    for k in ("k_pathtrace", "k_gbuffer", "k_gradient", "k_lut", "k_pair_weights"):
        if k in name:
            return k
    return None


def main(root):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row.get("Kernel_Name", ""))
                if not k:
                    continue
                # only the largest grid of each kernel (the 4K launches when several sizes ran)
                c = acc[k][row["Counter_Name"]]
                c[0] += float(row["Counter_Value"])
                c[1] += 1
    out = {}
    for k, cs in acc.items():
        out[k] = {n: v[0] / v[1] for n, v in cs.items()}
        out[k]["_dispatches"] = max(v[1] for v in cs.values())
        if "FETCH_SIZE" in out[k]:
            out[k]["fetch_bytes_raw"] = out[k]["FETCH_SIZE"] * 1024
            out[k]["fetch_bytes_corrected_x2"] = out[k]["FETCH_SIZE"] * 2048
        if "WRITE_SIZE" in out[k]:
            out[k]["write_bytes"] = out[k]["WRITE_SIZE"] * 1024
        if "fetch_bytes_corrected_x2" in out[k] and "write_bytes" in out[k]:
            out[k]["hbm_traffic_bytes"] = out[k]["fetch_bytes_corrected_x2"] + out[k]["write_bytes"]
        if "TCC_HIT_sum" in out[k]:
            out[k]["l2_hit_rate"] = out[k]["TCC_HIT_sum"] / max(1.0, out[k]["TCC_HIT_sum"] + out[k]["TCC_MISS_sum"])
        if "SQ_INSTS_VALU" in out[k] and "SQ_WAVES" in out[k]:
            out[k]["valu_insts_per_wave"] = out[k]["SQ_INSTS_VALU"] / max(1.0, out[k]["SQ_WAVES"])
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1])
