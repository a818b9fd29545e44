import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_torch as R
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _diaglib import use_env_library  # tools/_diaglib.py: DRAGPOSER_LIB names a diagnostic build

use_env_library()
from dragposer_amd.optimizer import LatentOptimizer
from dragposer_amd.drag_pose import DragPose
from test_temporal import _load_temporal
dev = torch.device("cuda:0"); opt = LatentOptimizer(device=dev)
for name in ("seq6", "seq3"):
    g = R.load_golden(os.path.join(ROOT, f"tests/golden/{name}.npz")); mt = g["meta"]; cfg = mt["cfg"]; K, T = mt["K"], mt["T"]
    dp = DragPose(opt, _load_temporal(g), g["means_latent"], g["stds_latent"], n_sequences=K)
    dp.set_initial_state(g["z0"], np.zeros((K, 3), np.float32), g["init_rot"], g["init_heights"])
    ja = tuple(cfg["joint_adjustment_indices"]) if cfg["enable_joint_adjustment"] else None
    for t in range(T):
        pose, gpos = dp.run(g["tgt_pos"][t], g["tgt_rot"][t], g["mask_idx"], g["weights"], stop_eps_pos=1e-4, stop_eps_rot=1e-2, max_iter=100,
                            min_loss_incr=1e-5, learning_rate=1e-2, lambda_rot=1, lambda_temporal=cfg["lambda_temporal"],
                            temporal_future_window=cfg["temporal_future_window"], joint_adjustment_indices=ja, joint_adjustment_weight=cfg["joint_adjustment_weight"])
        it = dp.last["iters"].cpu().numpy()
        print(name, t, "iters", it.tolist(), "ref", g["iters"][t].tolist(), "| gpos mm %.4f" % (np.abs(gpos.cpu().numpy() - g["gpos_ret"][t]).max() * 1000),
              "latent %.2e" % np.abs(dp.latent.cpu().numpy() - g["latent"][t]).max(), "rot %.2e" % np.abs(dp.current_global_rot.cpu().numpy() - g["cur_rot"][t]).max(),
              "ztgt %.2e" % np.abs(dp.target_latent_buffer[:, (dp.current_index - 1) % max(cfg["temporal_future_window"], 1)].cpu().numpy() - g["z_tgt"][t]).max(),
              "pose %.2e" % np.abs(pose.cpu().numpy()[:, 4:] - g["pose_ret"][t][:, 4:]).max(), flush=True)
