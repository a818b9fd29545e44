"""probe: teacher-forced per-frame comparison of the kernel with the reference's recorded sequences"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_torch as R
from dragposer_amd.optimizer import LatentOptimizer
dev = torch.device("cuda:0")
opt = LatentOptimizer(device=dev)
for name in ("seq6", "seq3"):
    g = R.load_golden(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
    cfg = g["meta"]["cfg"]; K, T = g["meta"]["K"], g["meta"]["T"]
    z_in = np.concatenate([g["z0"][None], g["latent"][:-1]], 0).reshape(T * K, 24)
    r_in = np.concatenate([g["init_rot"][None], g["cur_rot"][:-1]], 0).reshape(T * K, 4)
    idx = g["mask_idx"].astype(np.int64); E = len(idx)
    B = T * K
    tp = np.zeros((B, 22, 3), np.float32); tR = np.zeros((B, 22, 9), np.float32); w = np.zeros((B, 22, 2), np.float32); trk = np.zeros((B, 22), np.uint8)
    tp[:, idx] = g["tgt_pos"].reshape(B, E, 3); tR[:, idx] = g["tgt_rot"].reshape(B, E, 9); w[:, idx] = g["weights"]; trk[:, idx] = 1
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    o = opt.optimize(t(z_in), t(g["z_tgt"].reshape(B, 24)), t(r_in), t(tp), t(tR), t(w), t(trk), n_iter=100, lr=1e-2, lambda_rot=1.0,
                     lambda_tmp=float(cfg["lambda_temporal"]), stop_eps_pos=1e-4, stop_eps_rot=0.01, min_loss_incr=1e-5)
    o = {k: v.cpu().numpy() for k, v in o.items()}
    it_eq = o["iters"] == g["iters"].reshape(B)
    zerr = np.abs(o["z"] - g["latent"].reshape(B, 24)).max(1)
    rerr = np.abs(o["world_rot"] - g["cur_rot"].reshape(B, 4)).max(1)
    perr = np.abs(o["pose"][:, 4:] - g["pose_ret"].reshape(B, 88)[:, 4:]).max(1)
    print(name, "iters equal", it_eq.mean(), "mismatch at", np.nonzero(~it_eq)[0].tolist(), o["iters"][~it_eq].tolist(), g["iters"].reshape(B)[~it_eq].tolist())
    print("  z err     ", np.sort(zerr)[-6:], "median", np.median(zerr))
    print("  rot err   ", np.sort(rerr)[-6:], "median", np.median(rerr))
    print("  pose err  ", np.sort(perr)[-6:], "median", np.median(perr))
    print("  on equal-iter frames: z", zerr[it_eq].max(), "rot", rerr[it_eq].max(), "pose", perr[it_eq].max())
