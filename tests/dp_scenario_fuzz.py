"""Developer tool (not collected by pytest): two data-parallel ranks (gloo, both on GPU 0) through random sequences of steps
at CHANGING per-rank batch sizes (ragged ones included), fp32 and bf16 storage, with and without SyncBN and bf16 gradient
buckets.  The invariant of data parallelism: the replicas never diverge — after every sequence rank 0's and rank 1's
parameters, Adam moments and (with SyncBN) BatchNorm statistics are bit-identical, and the loss both ranks report is the
same number.

    python tests/dp_scenario_fuzz.py [sequences] [steps]"""
import os
import random
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BATCHES = [37, 64, 100, 385, 1024, 1236, 2048]


def worker(rank, world, port, out_dir, nseq, nsteps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bilinear_amd
        from bilinear_amd.dp import DataParallel
        dev = torch.device("cuda:0")
        for seq in range(nseq):
            rnd = random.Random(int(os.environ.get("FUZZ_SEED", "900")) + seq)                  # (the same draw on both ranks)
            dtype = rnd.choice(["fp32", "bf16s"])
            nb, width = rnd.choice([(1, 256), (2, 512), (2, 1024)])
            sync_bn = rnd.random() < 0.3
            compress = "bf16" if rnd.random() < 0.3 else None
            torch.manual_seed(seq)
            net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=width, gemm_dtype=dtype)
            net.train()
            net.engine.seed = 77
            dp = DataParallel(net, opt, bucket_floats=rnd.choice([None, 50000, 300000]), sync_bn=sync_bn, compress=compress)
            g = torch.Generator(device=dev).manual_seed(1000 * seq + rank)     # different data per rank
            losses = []
            for _ in range(nsteps):
                b = rnd.choice(BATCHES)
                x = torch.randn(b, 32, device=dev, generator=g)
                t = torch.randn(b, 48, device=dev, generator=g)
                if rnd.random() < 0.15:
                    opt.param_groups[0]["lr"] = rnd.choice([1e-3, 5e-4])
                losses.append(float(dp.train_step(x, t)[1].item()))
            torch.cuda.synchronize()
            np.save(os.path.join(out_dir, "p_%d_%d.npy" % (seq, rank)), net.engine.params.cpu().numpy())
            np.save(os.path.join(out_dir, "v_%d_%d.npy" % (seq, rank)), opt._exp_avg_sq.cpu().numpy())
            np.save(os.path.join(out_dir, "bn_%d_%d.npy" % (seq, rank)), net.engine.bn_running.cpu().numpy())
            np.save(os.path.join(out_dir, "l_%d_%d.npy" % (seq, rank)), np.array(losses))
            with open(os.path.join(out_dir, "cfg_%d.txt" % seq), "w") as f:
                f.write("%s %d x %d sync_bn=%s compress=%s" % (dtype, nb, width, sync_bn, compress))
    finally:
        dist.destroy_process_group()


def main():
    import tempfile
    nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = tempfile.mkdtemp()
    mp.spawn(worker, args=(2, port, out, nseq, nsteps), nprocs=2, join=True)
    bad = 0
    for seq in range(nseq):
        cfg = open(os.path.join(out, "cfg_%d.txt" % seq)).read()
        ld = lambda k, r: np.load(os.path.join(out, "%s_%d_%d.npy" % (k, seq, r)))
        same = np.array_equal(ld("p", 0), ld("p", 1)) and np.array_equal(ld("v", 0), ld("v", 1)) and \
            np.array_equal(ld("l", 0), ld("l", 1)) and np.isfinite(ld("p", 0)).all()
        if "sync_bn=True" in cfg:
            same = same and np.array_equal(ld("bn", 0), ld("bn", 1))
        bad += 0 if same else 1
        print("sequence %d (%s, %d steps): %s" % (seq, cfg, nsteps, "replicas bit-identical, losses agree" if same else "REPLICAS DIVERGED"),
              flush=True)
    print("%d sequences, %d diverged" % (nseq, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
