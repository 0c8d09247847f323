"""N>1 rehearsal of the drop-in class on ONE GPU: launch with
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 tools/rehearse_sharded_dropin.py
Both ranks share GPU 0 over a gloo group: the ARD grid cells and the candidates are sharded, the results must be
the reference's golden vectors on every rank."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from bayesian_optimisation_amd import PointSelector

torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for name in ("g1_m32", "g1_m50", "g2_n20_tr"):
    g = dict(np.load(os.path.join(root, name + ".npz")))
    ps = PointSelector()
    ps.name, ps.iteration = "T", 0
    ps.measured_pts, ps.measured_vals = g["X"], g["y"]
    ps.feature_domain = [int(v) for v in g["feature_domain"]]
    ps.predicted_pts, ps.length_scales = g["Xs"], g["length_scales"]
    ps.update_surrogate()
    idx = ps.lower_confidence_bound()
    ok = (np.array_equal(ps.kernel_params, g["kernel_params"]) and np.array_equal(idx, g["index"])
          and np.allclose(ps.nlogml, g["nlogml"], rtol=2e-6)
          and np.max(np.abs(ps.cov_func - g["cov_func"])) <= 1e-8 and ps.mean_func.shape == g["mean_func"].shape)
    print(f"rank {rank} {name}: kernel_params {np.ravel(ps.kernel_params).tolist()} index {idx.tolist()} ok={ok}", flush=True)
    assert ok
dist.barrier()
dist.destroy_process_group()
