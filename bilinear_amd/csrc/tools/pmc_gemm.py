#!/usr/bin/env python3
"""Developer tool: launch the forward GEMM of the W x W Linear in its three fp32-accurate forms of the W x W Linear a few times so that
`PYTHONPATH=. rocprofv3 --pmc ... -- python3 bilinear_amd/csrc/tools/pmc_gemm.py` can attribute counters to them; with `--sum DIR`
summarise a counter_collection.csv directory per kernel."""
import csv, ctypes, glob, sys, collections


def summarise(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gemm" not in n:
                continue
            n = n.replace("void blh::", "").split("(")[0][:70]
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
    B, W = 4096, 1024
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    A = torch.randn(B, W, device=dev); Wt = torch.randn(W, W, device=dev) * 0.03
    bias = torch.randn(W, device=dev); Z = torch.empty(B, W, device=dev)
    ws16 = torch.empty(lib.blh_gemm_fp16x2_workspace_bytes(), dtype=torch.uint8, device=dev)
    for i in range(12):
        N.check(lib.blh_gemm_fp16x2(st, A.data_ptr(), W, 0, Wt.data_ptr(), W, 0, Z.data_ptr(), W, B, W, W, 1,
                                    bias.data_ptr(), None, 0, ws16.data_ptr(), 1 if i else 0), "fp16x2 gemm")
        N.check(lib.blh_gemm_bf16x3(st, A.data_ptr(), W, 0, Wt.data_ptr(), W, 0, Z.data_ptr(), W, B, W, W, 1,
                                    bias.data_ptr(), None, 0), "split gemm")
        N.check(lib.blh_gemm_f32(st, A.data_ptr(), W, 0, Wt.data_ptr(), W, 0, Z.data_ptr(), W, B, W, W, 1,
                                 bias.data_ptr(), None, 0), "f32 gemm")
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
