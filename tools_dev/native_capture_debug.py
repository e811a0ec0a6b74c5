"""developer: where does the captured library-driven DP step die?  python -X faulthandler tools_dev/native_capture_debug.py <tail> <dtype>"""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import bilinear_amd
from bilinear_amd.dp import DataParallel, CapturedDataParallelStep
tail = sys.argv[1] if len(sys.argv) > 1 else "producer"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
torch.manual_seed(0)
net, opt, _, _ = bilinear_amd.load(dev, num_blocks=1, width=1024, gemm_dtype=dtype)
net.train()
B = 2048
x = torch.randn(B, 32, device=dev); t = torch.randn(B, 48, device=dev)
dp = DataParallel(net, opt, bucket_floats=200000, force_collectives=True, collectives="native", native_tail=tail)
print("eager step", flush=True)
print(float(dp.train_step(x, t)[1]), flush=True)
print("capture", flush=True)
cap = CapturedDataParallelStep(dp, B)
print("captured; replay", flush=True)
for i in range(3):
    print(float(cap(x, t)[1]), flush=True)
print("ok", flush=True)
dist.destroy_process_group()
