#!/usr/bin/env python3
"""summary.txt of tools_dev/pmc_hbm.sh -> profiles/r0N_hbm_traffic.json: HBM-side bytes per launch of the
HBM-bound kernels (skinny projections, BatchNorm kernels, clip + Adam and the small reductions) beside their
algorithmic bytes.  read bytes = FETCH_SIZE (KB) * 1024 * 2 (gfx950 tallies a 128-B request as 64 B on wide
streams: MI355X_MICROARCH.md, HBM), write bytes = WRITE_SIZE (KB) * 1024.  Narrow accesses (the 16-byte-per-row
operand loads of the 16x16 MFMA layouts, 4-byte partials) are NOT calibrated for the x2: for those kernels the
figure is an upper bound of the read side.  Infinity-Cache hits are counted by these counters (same section):
"traffic" is fabric traffic below the L2, not DRAM traffic."""
import json
import re
import sys

SHAPES = {1: dict(B=4096, W=1024, s=4, nh=5, P=4291632), 2: dict(B=16384, W=1024, s=2, nh=9, P=8498224),
          3: dict(B=8192, W=1024, s=2, nh=9, P=8498224), 4: dict(B=16384, W=2048, s=2, nh=17, P=67377200)}


def algorithmic(kernel, cfg):
    """(bytes per launch as the mean over the launches of one step, what they are) or (None, "")"""
    B, W, s, nh = cfg["B"], cfg["W"], cfg["s"], cfg["nh"]
    bw, bits = B * W * s, B * W // 8
    nskip = (nh - 1) // 2
    k = kernel
    if k.startswith("bn_apply_"):
        return bw * (2 + nskip / nh) + bits, "Z (+ skip on %d of %d stages) read, A + keep bits written" % (nskip, nh)
    if k.startswith("bn_bwd_reduce_"):
        return 2 * bw + bits, "dA, Z, keep bits read"
    if k.startswith("bn_bwd_apply_"):
        return 3 * bw + bits, "dA, Z, keep bits read, dZ written"
    if k.startswith("clip_adam_kernel"):
        return 28 * cfg["P"], "g, p, m, v read; p, m, v written (28 B per parameter)"
    if k.startswith("decode_fused_kernel"):
        return 4 * B * (2 * W + 3 * 48), "A, target read; pred, dpred, dA written (fp32)"
    if k.startswith("enc_fwd_kernel"):
        return 4 * B * (32 + W) + bits, "x read; A0 + keep-and-gate bits written"
    if k.startswith("enc_bwd_kernel"):
        return 4 * B * (W + 32) + bits, "dA0, bits, x read (+ 4.3 MB of row-block partials written)"
    if k.startswith("decode_fwd_mse_kernel"):
        return 4 * B * (W + 3 * 48), "A, target read; pred, dpred written (fp32)"
    if k.startswith("enc_fwd_h_kernel"):
        return 2 * B * (32 + W) + bits, "x (bf16) read; A0 (bf16) + keep-and-gate bits written"
    if k.startswith("enc_bwd_h_kernel"):
        return 2 * B * (W + 32) + bits, "dA0 (bf16), bits, x read (+ row-block partials written)"
    if k.startswith("decode_fwd_mse_h_kernel") and "true>" in k:
        return B * (2 * 2 * W + 4 * 48 * 3 + 2 * 48), "A (bf16), target read; pred, dpred (fp32 + bf16), dA (bf16) written"
    if k.startswith("decode_fwd_mse_h_kernel"):
        return B * (2 * W + 4 * 48 * 3 + 2 * 48), "A (bf16), target read; pred, dpred (fp32 + bf16) written"
    return None, ""


def main():
    cfg_id, rec, kernel = None, {}, None
    for line in open(sys.argv[1]):
        m = re.match(r"== config (\d+) pass (\w+)", line)
        if m:
            cfg_id = int(m.group(1))
            continue
        if not line.startswith(" "):
            kernel = line.strip()
            continue
        f = line.split()
        n = int(re.search(r"n=(\d+)", line).group(1))
        rec.setdefault(cfg_id, {}).setdefault(kernel, {})[f[0]] = (float(f[1]), n)
    out = {"note": " ".join(__doc__.split()), "configs": {}}
    for cid, kernels in sorted(rec.items()):
        cfg = SHAPES[cid]
        block = {"shape": cfg, "kernels": {}}
        for kernel, c in sorted(kernels.items()):
            if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                continue
            rd, wr = c["FETCH_SIZE"][0] * 2048, c["WRITE_SIZE"][0] * 1024
            e = {"launches_profiled": c["FETCH_SIZE"][1], "read_bytes": int(rd), "write_bytes": int(wr),
                 "traffic_bytes": int(rd + wr)}
            alg, what = algorithmic(kernel, cfg)
            if alg:
                e["algorithmic_bytes"] = int(alg)
                e["algorithmic_what"] = what
                e["traffic_over_algorithmic"] = round((rd + wr) / alg, 3)
            block["kernels"][kernel] = e
        out["configs"]["configs[%d]" % cid] = block
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
