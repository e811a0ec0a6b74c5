"""Developer: time the encode-stage kernels through their C entry points (fp32 and bf16 storage) with HIP events.
   python tools_dev/enc_time.py [B W] ..."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bilinear_amd import _native as N

def time_it(fn, reps=300, warm=30):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return 1e3 * a.elapsed_time(b) / reps

def run(B, W):
    lib = N.lib(); dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    drop = N.Dropout(None, 1, 0, 0, 0, 0)
    gam, bet = torch.ones(W, device=dev), torch.zeros(W, device=dev)
    rm, rv = torch.zeros(W, device=dev), torch.ones(W, device=dev)
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    saved = torch.empty(4, W, device=dev)
    b0 = torch.randn(W, device=dev)
    dW0 = torch.empty(W, 32, device=dev); db0, dg, dbe = (torch.empty(W, device=dev) for _ in range(3))
    out = []
    # fp32
    x = torch.randn(B, 32, device=dev); W0 = torch.randn(W, 32, device=dev) * 0.25
    A = torch.empty(B, W, device=dev); dA = torch.randn(B, W, device=dev)
    bits = torch.zeros(((B + 7) // 8) * (W // 4), dtype=torch.int32, device=dev)
    scratch = torch.empty(B * W, device=dev)
    f = lambda: lib.blh_skinny_encode_fused_fwd(st, x.data_ptr(), W0.data_ptr(), b0.data_ptr(), gam.data_ptr(), bet.data_ptr(),
        rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1, saved.data_ptr(), scratch.data_ptr(), A.data_ptr(), bits.data_ptr(),
        ctypes.byref(drop), B, W, 32)
    g = lambda: lib.blh_skinny_encode_fused_bwd(st, dA.data_ptr(), x.data_ptr(), W0.data_ptr(), b0.data_ptr(), saved.data_ptr(),
        bits.data_ptr(), scratch.data_ptr(), dW0.data_ptr(), db0.data_ptr(), dg.data_ptr(), dbe.data_ptr(), B, W, 32)
    if f() == 0:
        out.append("fp32 fwd %.1f us bwd %.1f us" % (time_it(f), time_it(g)))
    del A, dA, scratch
    # bf16
    xh = x.to(torch.bfloat16); W0h = W0.to(torch.bfloat16)
    Ah = torch.empty(B, W, dtype=torch.bfloat16, device=dev); dAh = torch.randn(B, W, device=dev).to(torch.bfloat16)
    bith = torch.zeros(((B + 3) // 4) * (W // 8), dtype=torch.int32, device=dev)
    scr = torch.empty(B * W, dtype=torch.bfloat16, device=dev)
    fh = lambda: lib.blh_skinny_encode_fused_fwd_bf16(st, xh.data_ptr(), W0h.data_ptr(), b0.data_ptr(), gam.data_ptr(), bet.data_ptr(),
        rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1, saved.data_ptr(), scr.data_ptr(), Ah.data_ptr(), bith.data_ptr(),
        ctypes.byref(drop), B, W, 32)
    gh = lambda: lib.blh_skinny_encode_fused_bwd_bf16(st, dAh.data_ptr(), xh.data_ptr(), W0h.data_ptr(), b0.data_ptr(), saved.data_ptr(),
        bith.data_ptr(), scr.data_ptr(), dW0.data_ptr(), db0.data_ptr(), dg.data_ptr(), dbe.data_ptr(), B, W, 32)
    if fh() == 0:
        out.append("bf16 fwd %.1f us bwd %.1f us" % (time_it(fh), time_it(gh)))
    print("B %6d W %5d: %s" % (B, W, " | ".join(out)), flush=True)

if __name__ == "__main__":
    shapes = [(4096, 1024), (16384, 1024), (8192, 1024), (16384, 2048)]
    a = sys.argv[1:]
    if a: shapes = [(int(a[i]), int(a[i + 1])) for i in range(0, len(a), 2)]
    run(4096, 1024)   # clock ramp
    for B, W in shapes: run(B, W)
