"""Where a device optimisation and the CPU restatement part ways (run through gpurun): the fuzz scene of
tests/test_gpu_optimizer.py::test_randomised_optimiser_against_cpu[seed], maximize! with max_iters = 1 .. N on both.
usage: gpu_optim_trace_seed.py seed [seed ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import celeste_jl_amd as cel
from celeste_jl_amd import synthetic
from oracle import oracle

for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(5000 + seed)
    S = int(rng.integers(2, 9))
    f = synthetic.make_field(int(rng.integers(60, 100)), int(rng.integers(60, 100)), S, seed=6000 + seed,
                             nan_fraction=float(rng.choice([0.0, 0.01])), margin=int(rng.integers(10, 25)))
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
    tg = rng.permutation(S)[:int(rng.integers(1, min(S, 4) + 1))].tolist()
    iters = int(rng.choice([3, 8, 20]))
    lw = float(rng.choice([1e-4, 1.0]))
    print("seed %d: S %d targets %s iters %d loc_width %g" % (seed, S, tg, iters, lw))
    for k, t in enumerate(tg):
        prev = None
        for n in range(1, iters + 1):
            vp, its, evals, elbo, st = ctx.maximize_batch(f.vp, [t], cel.ElboConfig(max_iters=n, loc_width=lw))
            ovp, oit, oev, oelbo, ost = oracle.maximize(ctx.problem, f.vp, t, oracle.OptCfg(max_iters=n, loc_width=lw))
            d = abs(elbo[0] - oelbo) / abs(oelbo)
            print("  target %d max_iters %2d: device its %2d evals %2d elbo %.10f | cpu its %2d evals %2d elbo %.10f | rel diff %.1e  max|dvp| %.1e"
                  % (t, n, its[0], evals[0], elbo[0], oit, oev, oelbo, d, np.abs(vp[t] - ovp[t]).max()))
