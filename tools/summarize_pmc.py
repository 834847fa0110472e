"""Condense the rocprofv3 --pmc passes of tools/gpu_round.sh <tag> into one JSON for profiles/:
per kernel (template instantiation) mean HBM bytes per launch and MFMA utilisation.
usage: python tools/summarize_pmc.py <tag> [gpurun_out] > profiles/<tag>_kernels_pmc.json"""
import collections
import csv
import glob
import json
import os
import re
import sys

PAT = re.compile(r"(gemm_nt_kernel<[^>]*>|gemm_tn_kernel<[^>]*>|gemm_nt8_kernel<[^>]*>|gemm_tn8_kernel<[^>]*>|attn_rpe2d_\w+_kernel|ln_\w+_kernel|adamw_mirror_kernel|"
                 r"grad_finalize_kernel|rpe_gather_planes?<[^>]*>|rpe_scatter_planes|irpe_attn_\w+_kernel<[^>]*>|irpe_table_grad_kernel)")
EPI = {"0": "store", "1": "bias", "2": "bias+gelu (two outputs)", "3": "x gelu' + column sums"}
def _epi(k):
    """epilogue template argument of an NT kernel name: 6th of gemm_nt_kernel<...>, 1st of gemm_nt8_kernel<...>"""
    p = [x.strip() for x in k[k.index("<") + 1:-1].split(",")]
    return p[0] if k.startswith("gemm_nt8_kernel") else p[5]


FAMILY = {"gemm_nt": lambda k: k.startswith(("gemm_nt_kernel", "gemm_nt8_kernel")) and _epi(k) in ("0", "1"),
          "gemm_nt_gelu": lambda k: k.startswith(("gemm_nt_kernel", "gemm_nt8_kernel")) and _epi(k) == "2",
          "gemm_nt_mul": lambda k: k.startswith(("gemm_nt_kernel", "gemm_nt8_kernel")) and _epi(k) == "3",
          "gemm_tn_wgrad": lambda k: k.startswith(("gemm_tn_kernel", "gemm_tn8_kernel")),
          "attn_rpe2d_fwd": lambda k: k in ("attn_rpe2d_fwd_kernel", "attn_rpe2d_fwd14_kernel"),
          # the one-pass backward (round 4) is ONE kernel per launch of the region; the two-launch pair only when it is absent
          "attn_rpe2d_bwd": lambda k: k == "attn_rpe2d_bwd1_kernel",
          "attn_rpe2d_bwd_pair": lambda k: k in ("attn_rpe2d_bwd_q_kernel", "attn_rpe2d_bwd_kv_kernel")}


def main():
    tag = sys.argv[1]
    root = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(root, f"{tag}_pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                m = PAT.search(r["Kernel_Name"])
                if m:
                    vals[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for k, d in sorted(vals.items()):
        mean = {c: sum(v) / len(v) for c, v in d.items()}
        rec = dict(launches_sampled=max(len(v) for v in d.values()))
        if k.startswith("gemm_nt_kernel"):
            p = [x.strip() for x in k[k.index("<") + 1:-1].split(",")]
            rec["tile"], rec["epilogue"] = f"{p[0]}x{p[1]}", EPI.get(p[5], p[5])
        if k.startswith("gemm_nt8_kernel"):
            rec["tile"], rec["epilogue"] = "256x256 (phase-interleaved loop)", EPI.get(_epi(k), _epi(k))
        if k.startswith("gemm_tn8_kernel"):
            rec["tile"] = "256x256 (phase-interleaved loop, bf16 partial tiles)"
        if "FETCH_SIZE" in mean:
            rec["hbm_read_MB_corrected"] = round(mean["FETCH_SIZE"] * 2 / 1024, 1)
        if "WRITE_SIZE" in mean:
            rec["hbm_write_MB"] = round(mean["WRITE_SIZE"] / 1024, 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and mean.get("GRBM_GUI_ACTIVE"):
            rec["mfma_util"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (mean["GRBM_GUI_ACTIVE"] / 8 * 256 * 4), 4)
        kernels[k] = rec
    fam = {}
    for name, pred in FAMILY.items():
        tot_b, tot_n = 0.0, 0
        for k, rec in kernels.items():
            if pred(k) and "hbm_read_MB_corrected" in rec:
                n = rec["launches_sampled"]
                tot_b += (rec["hbm_read_MB_corrected"] + rec.get("hbm_write_MB", 0.0)) * 1e6 * n
                tot_n += n
        if tot_n:
            # the two-launch attention backward = two kernels per launch of the region
            per = 2 if name == "attn_rpe2d_bwd_pair" else 1
            fam[name] = int(tot_b / tot_n * per)
    if "attn_rpe2d_bwd" not in fam and "attn_rpe2d_bwd_pair" in fam:
        fam["attn_rpe2d_bwd"] = fam["attn_rpe2d_bwd_pair"]
    out = dict(source=f"rocprofv3 --kernel-trace --pmc <group> -- python bench.py --steps 2 --warmup 1 (tools/gpu_round.sh {tag}), "
                      "one pass per counter group, means over the sampled launches (the sub-network changes per step)",
               units=dict(hbm_read_MB_corrected="2 x FETCH_SIZE[KiB] / 1024 (gfx950: wide coalesced reads are tallied at half "
                                                "their bytes, MI355X_MICROARCH.md HBM section)",
                          hbm_write_MB="WRITE_SIZE[KiB] / 1024",
                          mfma_util="SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)"),
               traffic_bytes_per_launch_by_timed_region=fam, kernels=kernels)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
