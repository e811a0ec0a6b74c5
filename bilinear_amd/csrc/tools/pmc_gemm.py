#!/usr/bin/env python3
"""Developer tool: launch the hidden-layer GEMMs exactly as the library ships them (fp32 ring
kernel: forward with BatchNorm partials, dgrad, wgrad; bf16-storage kernel: forward) a few times,
so that

    PYTHONPATH=. rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- \
        python3 bilinear_amd/csrc/tools/pmc_gemm.py [B [W]]

can attribute counters to them; `pmc_gemm.py --sum <dir>` prints the mean per launch of every
counter per kernel (summed over the chip)."""
import collections
import csv
import ctypes
import glob
import sys


def summarise(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gemm" not in n:
                continue
            n = n.replace("void blh::", "").split("(")[0][:80]
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n in sorted(acc):
        print(n)
        for c in sorted(acc[n]):
            v = acc[n][c]
            print("   %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--sum":
        return summarise(sys.argv[2])
    import torch
    from bilinear_amd import _native as N
    lib = N.lib()
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    A = torch.randn(B, W, device=dev)
    Wt = torch.randn(W, W, device=dev) * 0.03
    bias = torch.randn(W, device=dev)
    Z = torch.empty(B, W, device=dev)
    stat = torch.empty((B + 127) // 128, 2, W, device=dev)
    splits = max(1, min((256 * 128 * 128) // (W * W), B // 128))
    slabs = torch.empty(splits, W, W, device=dev)
    Ah, Wh = A.to(torch.bfloat16), Wt.to(torch.bfloat16)
    Zh = torch.empty(B, W, dtype=torch.bfloat16, device=dev)
    Gh = torch.randn(B, W, device=dev).to(torch.bfloat16)
    # weight gradient in bf16 storage: the batch slabs api_layout.h (wgrad_plan_h) picks
    t256 = (W // 256) * (W // 256)
    if t256 >= 64 and B % (max(1, 256 // t256) * 128) == 0:
        hsplits = max(1, 256 // t256)
    else:
        hsplits = max(1, min((256 * 128 * 128) // (W * W), B // 128))
    hslabs = torch.empty(hsplits, W, W, device=dev)
    only_h = len(sys.argv) > 3 and sys.argv[3] == "bf16s"
    for _ in range(12):
        N.check(lib.blh_gemm_bf16s(st, Ah.data_ptr(), W, 0, Wh.data_ptr(), W, 0, Zh.data_ptr(), W, 1, B, W, W, 1,
                                   bias.data_ptr(), None, 0, stat.data_ptr()), "bf16s fwd")
        N.check(lib.blh_gemm_bf16s(st, Gh.data_ptr(), W, 0, Wh.data_ptr(), W, 1, Zh.data_ptr(), W, 1, B, W, W, 1,
                                   None, None, 0, None), "bf16s dgrad")
        N.check(lib.blh_gemm_bf16s(st, Gh.data_ptr(), W, 1, Ah.data_ptr(), W, 1, hslabs.data_ptr(), W, 0, W, W, B,
                                   hsplits, None, None, 0, None), "bf16s wgrad")
        if only_h:
            continue
        N.check(lib.blh_linear_fwd_stats(st, A.data_ptr(), Wt.data_ptr(), bias.data_ptr(), Z.data_ptr(),
                                         stat.data_ptr(), B, W, W), "fwd")
        N.check(lib.blh_gemm_f32(st, A.data_ptr(), W, 0, Wt.data_ptr(), W, 1, Z.data_ptr(), W, B, W, W, 1,
                                 None, None, 0), "dgrad")
        N.check(lib.blh_gemm_f32(st, A.data_ptr(), W, 1, Z.data_ptr(), W, 1, slabs.data_ptr(), W, W, W, B,
                                 splits, None, None, 0), "wgrad")
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
