import numpy as np, torch, bilinear_amd
dev = torch.device("cuda", 0)
def make(small):
    torch.manual_seed(11)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=2, width=1024)
    net.train(); net.engine.ensure(dev); net.engine.seed = 4242; net.engine.set_small_step(small)
    return net, opt
masks = (np.random.default_rng(0).random((5, 64, 1024)) < 0.5).astype(np.uint8)
g = torch.Generator().manual_seed(9)
x = torch.randn(64, 32, generator=g).to(dev); t = torch.randn(64, 48, generator=g).to(dev)
crit = torch.nn.MSELoss()
def run(net, opt, sync=False):
    net.engine.set_dropout_masks(masks)
    opt.zero_grad(); loss = crit(net(x), t)
    if sync: torch.cuda.synchronize()
    loss.backward(); torch.cuda.synchronize()
    return net.engine.grads.clone()
nm, om = make(False); gm = run(nm, om)
na, oa = make(True)
for k in range(4):
    ga = run(na, oa, sync=(k >= 2))
    d = (ga - gm).double().norm() / gm.double().norm()
    print("run %d (sync between=%s): rel vs multi %.3e" % (k, k >= 2, float(d)))
    if k: print("   vs previous run: equal=%s" % torch.equal(ga, prev))
    prev = ga
