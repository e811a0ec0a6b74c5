#!/usr/bin/env python3
"""summary.txt of tools_dev/pmc_r04.sh -> the traffic record bench.py reads (profiles/r04_traffic.json).
read bytes = FETCH_SIZE (KB) * 1024 * 2 (gfx950 counts 64 B per 128-B request on wide streams,
MI355X_MICROARCH.md, HBM), write bytes = WRITE_SIZE (KB) * 1024."""
import json
import re
import sys


def main():
    shape, rec = None, {}
    for line in open(sys.argv[1]):
        m = re.match(r"== M=(\d+) W=(\d+) pass", line)
        if m:
            shape = "%sx%s" % (m.group(1), m.group(2))
            continue
        if not line.startswith(" "):
            kernel = line.strip()
            continue
        name, val = line.split()[0], float(line.split()[1])
        rec.setdefault(shape, {}).setdefault(kernel, {})[name] = val
    out = {"note": "HBM-side traffic and matrix-pipe busy of the shipped hidden-layer GEMMs, rocprofv3 PMC in separate "
                   "passes (--pmc FETCH_SIZE | WRITE_SIZE | SQ counters, each with --kernel-trace only) around "
                   "bilinear_amd/csrc/tools/pmc_gemm.py on one MI355X (tools_dev/pmc_r04.sh); bytes per launch, mean of "
                   "12 launches; read bytes = FETCH_SIZE(KB)*1024*2 (gfx950 counts 64 B per 128-B request on wide "
                   "streams), write bytes = WRITE_SIZE(KB)*1024; busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * "
                   "SQ_WAVE_CYCLES / waves per SIMD), two waves per SIMD resident in the big-tile kernels",
           "shapes": {}}
    for shape, kernels in rec.items():
        M, W = (int(v) for v in shape.split("x"))
        for kernel, c in kernels.items():
            if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                continue
            bf16 = "bf16s" in kernel
            es = 2 if bf16 else 4
            wgrad = "<1, 1" in kernel or "1, 1, 0" in kernel.split("<")[1][:30] and not bf16
            alg = es * (2 * M * W + W * W) if not wgrad else es * 2 * M * W + 4 * W * W
            e = {"fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
                 "read_bytes": int(c["FETCH_SIZE"] * 2048), "write_bytes": int(c["WRITE_SIZE"] * 1024),
                 "traffic_bytes": int(c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024),
                 "algorithmic_bytes_one_slab": alg}
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_WAVE_CYCLES"):
                e["mfma_busy_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]
                e["wave_quad_cycles"] = c["SQ_WAVE_CYCLES"]
                e["matrix_pipe_busy_2_waves_per_simd"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * c["SQ_WAVE_CYCLES"] / 2)
                e["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
            out["shapes"].setdefault(shape, {})[kernel] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
