import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import bench
import celeste_jl_amd as cel
fld = bench.build_field(2048, 1489, 2000, 3)
ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
_, its, evals, _, st = ctx.maximize_batch(fld.vp, np.arange(2000, dtype=np.int32), cel.ElboConfig())
live = [int((evals > k).sum()) for k in range(52)]
print("live targets at evaluation k:", live)
