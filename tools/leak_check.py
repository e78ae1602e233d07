import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import abstractgps_jl_amd as agp
from _synth import synth_inputs
x, y = synth_inputs(3000, 3, 1)
f = agp.GP(agp.SqExponentialKernel() @ agp.ScaleTransform(0.9))
def once():
    fx = f(agp.RowVecs(x), 0.05)
    p = agp.posterior(fx, y); p.mean_and_var(agp.RowVecs(x[:300])); p.cov(agp.RowVecs(x[:200]))
    p2 = agp.posterior(p(agp.RowVecs(x[:500] + 0.01), 0.05), y[:500])
    agp.logpdf(fx, np.stack([y, y], 1)); agp.logpdf_and_grad(fx, y); agp.rand(fx, 2, xi=np.ones((3000, 2)))
    z = x[:256]
    v = agp.posterior(agp.VFE(f(agp.RowVecs(z), 1e-6)), fx, y); v.mean_and_var(agp.RowVecs(x[:100])); agp.elbo(agp.VFE(f(agp.RowVecs(z), 1e-6)), fx, y)
    x32 = x.astype(np.float32); agp.elbo(agp.VFE(f(agp.RowVecs(z.astype(np.float32)), 1e-4)), f(agp.RowVecs(x32), np.float32(0.05)), y.astype(np.float32))
for i in range(3): once()
import gc; gc.collect(); torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
for i in range(40): once()
gc.collect(); torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("free before %.1f MB after %.1f MB delta %.1f MB" % (free0/2**20, free1/2**20, (free0-free1)/2**20))
