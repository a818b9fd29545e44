"""`eval_drag` on MI355X: the reference's offline evaluation CLI (python/src/eval_drag.py:21-293) with the
per-frame optimisation running in the HIP kernel.

    python -m dragposer_amd.eval_drag models/model_dancedb data/example/eval/example.bvh --config config/6_trackers_config.json

The reference's contract: `model_path` is the model FOLDER (generator.pt, data.pt and -- when trained -- temporal.pt, as
train.py / train_temporal.py save them), `input_path` a .bvh file or a directory of them, --config / --verbose as there, the
result written to data/eval_<name> relative to the working directory, and the reference's four result lines printed
verbatim (eval_drag.py:249-252).  Also accepted as `model_path`: this package's flat .npz fixture (default: the shipped
model_dancedb).  Stated in the output: because the reference's temporal.pt is not distributed with it the temporal
predictor is optional -- without one (no temporal.pt in the folder, no --temporal-checkpoint) the pull term is switched off
(lambda_temporal = 0) instead of pulling towards the predictions of an untrained network.
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from . import quat_np as Q
from .bvh import BVH
from .drag_pose import DragPose
from .encoder import PoseEncoder
from .model import DEFAULT_MODEL, NJ, load_model_arrays
from .motion import local_quats_from_bvh, prepare_motion
from .optimizer import LatentOptimizer
from .temporal import load_reference_checkpoint

SPARSE_JOINTS = [0, 4, 8, 13, 17, 21]  # train.py:32-39 (evaluation only)
HEIGHT_INDICES = [0, 4, 8, 13, 17, 21]

DEFAULT_CONFIG = dict(  # eval_drag.py:68-131
    mask=[1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1],
    weights=[[10, 10]] + [[1, 0.01]] * 2 + [[5, 0.01]] + [[1, 0.01]] * 3 + [[5, 0.01]] + [[1, 0.01]] * 5 + [[5, 0.01]]
    + [[1, 0.01]] * 3 + [[5, 0.01]] + [[1, 0.01]] * 3 + [[5, 0.01]],
    enable_joint_adjustment=True, joint_adjustment_indices=[0, 0], joint_adjustment_weight=1.0,
    lambda_temporal=0.02, temporal_future_window=0)


def eval_pos_error(gt_bvh, eval_bvh):
    """eval_metrics.eval_pos_error (eval_metrics.py:6-32): FK of both files with the root at the origin."""
    gq, _, parents, offsets = local_quats_from_bvh(gt_bvh)
    eq, _, _, _ = local_quats_from_bvh(eval_bvh)
    n = min(len(gq), len(eq))
    gp, _ = Q.fk(gq[:n], np.zeros((n, 3)), offsets, parents)
    ep, _ = Q.fk(eq[:n], np.zeros((n, 3)), offsets, parents)
    err = np.linalg.norm(ep - gp, axis=-1)
    return float(err.mean()), float(err[:, SPARSE_JOINTS[1:]].mean())


def result_to_bvh(poses, global_pos, means, stds, bvh, out_path):
    """train.result_to_bvh with are_root_rot_incr=False (train.py:437-509): normalised root-space quaternions
    whose root channels already hold the world rotation -> local Euler angles -> BVH."""
    sd4 = stds["dqs"].reshape(NJ, 8)[:, :4].reshape(88)
    mu4 = means["dqs"].reshape(NJ, 8)[:, :4].reshape(88)
    qs = (poses.astype(np.float64) * sd4 + mu4).reshape(-1, NJ, 4)
    _, _, parents, _, order = bvh.get_data()
    local = Q.from_root_space(qs, parents)
    rot = np.stack([np.degrees(Q.to_euler(local[:, j], order[j])) for j in range(NJ)], axis=1)
    bvh.set_data(rot, global_pos.astype(np.float64))
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    bvh.save(out_path)


def synthesize_targets(m, n_frames, mask_idx):
    """The per-frame targets of eval_drag.py:164-202 that do not depend on the running state: ground-truth root-space quaternions
    with the root channel replaced by the global rotation -> parent-local (from_root_quat) -> FK with the root at the origin.
    -> positions of the tracked joints relative to the root [F, E, 3] (the reference adds target_global_pos - current_global_pos
    to them, frame by frame: run_sequences does that on the device) and their global rotation matrices [F, E, 9]."""
    rq = m["root_quats"].copy()
    rq[:, 0] = m["global_rot"]
    local = Q.from_root_space(rq, m["parents"])
    p_rel, g_rot = Q.fk(local[:n_frames], np.zeros((n_frames, 3)), m["offsets"].astype(np.float64), m["parents"])
    return p_rel[:, mask_idx], Q.to_matrix(g_rot[:, mask_idx]).reshape(n_frames, len(mask_idx), 9)


def prepare_file(args, input_path, opt, encoder, cfg, raw):
    """BVH -> everything one sequence needs on the device (eval_drag.py:133-202): normalised motion, per-frame
    targets relative to the root, the ground-truth root trajectory, the encoder's initial latent."""
    dev = opt.device
    means = {"dqs": raw["means.dqs"], "displacement": raw["means.displacement"]}
    stds = {"dqs": raw["stds.dqs"], "displacement": raw["stds.displacement"]}
    bvh = BVH().load(input_path)
    m = prepare_motion(bvh, means, stds, HEIGHT_INDICES)
    if not np.allclose(m["offsets"], opt.host_model.arrays["offsets"], atol=1e-5) or list(m["parents"]) != list(opt.host_model.parents):
        raise SystemExit(f"{input_path}: skeleton differs from the one the model fixture was exported with")
    n_frames = len(m["dqs"]) if args.max_frames is None else min(args.max_frames, len(m["dqs"]))
    mask_idx = np.nonzero(np.asarray(cfg["mask"]))[0]
    # targets (eval_drag.py:164-202): FK of the ground-truth pose with the root at the origin, per frame; the
    # root translation relative to the running estimate is added inside the loop, on the device
    p_rel, r_mats = synthesize_targets(m, n_frames, mask_idx)
    if getattr(args, "initial_latent", None):
        # (parity runs: the reference draws the encoder's noise from torch's GLOBAL generator after its model constructors have
        #  consumed an implementation-defined amount of it -- eval_drag.py:23,49-51,152 -- so its latent is handed over instead)
        z0 = torch.tensor(np.load(args.initial_latent).reshape(1, -1), dtype=torch.float32, device=dev)
    else:
        gen = torch.Generator(device="cpu").manual_seed(2222)  # train.param["seed"]
        z0 = encoder.sample(torch.tensor(m["dqs"][0:1], device=dev), generator=gen)  # drag_pose.py:50
    return dict(path=input_path, bvh=bvh, m=m, n_frames=n_frames, means=means, stds=stds, z0=z0,
                tp_rel=torch.tensor(p_rel, dtype=torch.float32, device=dev), tR=torch.tensor(r_mats, dtype=torch.float32, device=dev),
                gpos=torch.tensor(m["global_pos"][:n_frames], dtype=torch.float32, device=dev))


def run_sequences(args, seqs, opt, temporal_pack, cfg):
    """The frame loop (eval_drag.py:181-227) for S prepared sequences advancing in lock-step: one `DragPose.run`
    (two kernel launches) per frame index whatever S.  Sequences shorter than the longest keep receiving their last
    frame's targets; their surplus frames are dropped."""
    dev, S = opt.device, len(seqs)
    T = max(q["n_frames"] for q in seqs)
    mask_idx = np.nonzero(np.asarray(cfg["mask"]))[0]
    weights = np.asarray(cfg["weights"], np.float32)[mask_idx]
    temporal, means_latent, stds_latent = temporal_pack
    lam_tmp = cfg["lambda_temporal"] if temporal is not None else 0.0
    window = cfg["temporal_future_window"] if temporal is not None else 0
    pad = lambda t: torch.cat((t, t[-1:].expand(T - t.shape[0], *t.shape[1:])), dim=0) if t.shape[0] < T else t
    tp_rel = torch.stack([pad(q["tp_rel"]) for q in seqs], dim=1).contiguous()  # [T, S, E, 3]
    tR = torch.stack([pad(q["tR"]) for q in seqs], dim=1).contiguous()
    gpos = torch.stack([pad(q["gpos"]) for q in seqs], dim=1).contiguous()      # [T, S, 3]

    drag = DragPose(opt, temporal, means_latent, stds_latent, n_sequences=S, native_temporal=not getattr(args, "torch_temporal", False))
    drag.set_initial_state(torch.cat([q["z0"] for q in seqs], dim=0), np.stack([q["m"]["global_pos"][0] for q in seqs]),
                           np.stack([q["m"]["global_rot"][0] for q in seqs]), np.stack([q["m"]["heights"][0] for q in seqs]))
    ja = tuple(cfg["joint_adjustment_indices"]) if cfg["enable_joint_adjustment"] else None
    poses = torch.zeros(T, S, 88, device=dev)
    out_pos = torch.zeros(T, S, 3, device=dev)
    iters = torch.zeros(T, S, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    if not getattr(args, "per_frame", False):
        # the frame loop on the device: one launch per stretch of frames between two temporal predictions (DragPose.run_frames)
        print(f"Frames: {T} (frame loop on the device)", flush=True)
        poses, out_pos, iters = drag.run_frames(tp_rel, tR.reshape(T, S, -1, 3, 3), mask_idx, weights, target_root=gpos, stop_eps_pos=0.01 * 0.01,
                                                stop_eps_rot=0.01, max_iter=args.max_iter, min_loss_incr=0.00001, learning_rate=1e-2, lambda_rot=1,
                                                lambda_temporal=lam_tmp, temporal_future_window=window, height_indices=HEIGHT_INDICES,
                                                joint_adjustment_indices=ja, joint_adjustment_weight=cfg["joint_adjustment_weight"])
    for i in range(T if getattr(args, "per_frame", False) else 0):
        if i % 1000 == 0:
            print(f"Frame: {i + 1} out of {T}", flush=True)
        tp = tp_rel[i] + (gpos[i] - drag.current_global_pos).unsqueeze(1)  # eval_drag.py:186-199
        drag.run(tp, tR[i], mask_idx, weights, stop_eps_pos=0.01 * 0.01, stop_eps_rot=0.01, max_iter=args.max_iter,
                 min_loss_incr=0.00001, learning_rate=1e-2, lambda_rot=1, lambda_temporal=lam_tmp,
                 temporal_future_window=window, height_indices=HEIGHT_INDICES, joint_adjustment_indices=ja,
                 joint_adjustment_weight=cfg["joint_adjustment_weight"], verbose=args.verbose, out_pose=poses[i], out_pos=out_pos[i])
        iters[i] = drag.last["iters"]
    torch.cuda.synchronize()
    elapsed = time.time() - t0
    poses, out_pos, iters = poses.cpu().numpy(), out_pos.cpu().numpy(), iters.cpu().numpy()
    return [dict(poses=poses[:q["n_frames"], k], pos=out_pos[:q["n_frames"], k], iters=iters[:q["n_frames"], k]) for k, q in enumerate(seqs)], elapsed, lam_tmp, temporal is not None


def finish_file(args, q, res, elapsed, lam_tmp, has_temporal, shared=1):
    """write the result BVH, report the reference's metrics (eval_drag.py:229-252)"""
    name = os.path.basename(q["path"])
    out_path = os.path.join(args.out_dir, "eval_" + name)
    result_to_bvh(res["poses"], res["pos"], q["means"], q["stds"], q["bvh"], out_path)
    mpjpe, mpeepe = eval_pos_error(BVH().load(q["path"]), BVH().load(out_path))
    n = q["n_frames"]
    print(f"Evaluate Loss: {mpjpe + mpeepe}")
    print(f"Mean Per Joint Position Error: {mpjpe}")
    print(f"Mean End Effector Position Error: {mpeepe}")
    print(f"Time: {elapsed}" + (f"  (shared by {shared} sequences in lock-step)" if shared > 1 else ""))
    print(f"Frames: {n}  ({n / elapsed:.1f} frames/s, mean iterations/frame {res['iters'].mean():.1f}, "
          f"lambda_temporal {lam_tmp}{'' if has_temporal else ' -- no temporal checkpoint given: pull term off'})")
    out = dict(mpjpe=mpjpe, mpeepe=mpeepe, time=elapsed, frames=n, out=out_path, mean_iters=float(res["iters"].mean()))
    if getattr(args, "keep_frames", False):
        out.update(poses=res["poses"], pos=res["pos"], iters=res["iters"])
    return out


def evaluate_files(args, paths, opt, encoder, temporal_pack, cfg, raw):
    """one or more files as sequences in lock-step"""
    seqs = [prepare_file(args, p, opt, encoder, cfg, raw) for p in paths]
    results, elapsed, lam_tmp, has_temporal = run_sequences(args, seqs, opt, temporal_pack, cfg)
    out = []
    for q, r in zip(seqs, results):
        if len(seqs) > 1:
            print(f"Result {q['path']} ------------------------")
        out.append(finish_file(args, q, r, elapsed, lam_tmp, has_temporal, shared=len(seqs)))
    return out


def evaluate_file(args, input_path, opt, encoder, temporal_pack, cfg, raw):
    return evaluate_files(args, [input_path], opt, encoder, temporal_pack, cfg, raw)[0]


def main(argv=None):
    ap = argparse.ArgumentParser(description="Evaluate DragPoser (HIP backend)")
    ap.add_argument("model_path", nargs="?", default=DEFAULT_MODEL,
                    help="path to pytorch model folder (generator.pt, data.pt[, temporal.pt]) as the reference takes it, or a model fixture (.npz); "
                         "default: the shipped model_dancedb")
    ap.add_argument("input_path", help=".bvh file or a directory of .bvh files")
    ap.add_argument("--config", default=None, help="tracker config JSON (same keys as the reference's config/*.json)")
    ap.add_argument("--temporal-checkpoint", default=None, help="temporal.pt as saved by the reference's train_temporal.py")
    ap.add_argument("--torch-temporal", action="store_true",
                    help="run the temporal target block with PyTorch ops around nn.Transformer instead of dp_temporal_predict (one HIP launch)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--max-iter", type=int, default=100)
    ap.add_argument("--max-frames", type=int, default=None)
    ap.add_argument("--out-dir", default="data")
    ap.add_argument("--keep-frames", action="store_true", help="callers of main(): also return the per-frame poses / global positions / iteration counts")
    ap.add_argument("--initial-latent", default=None,
                    help=".npy with the 24 numbers of the initial latent (instead of encoder(mu, logvar) + a seeded normal draw): for parity runs against "
                         "a recorded reference run, whose own draw depends on how much of torch's global generator its constructors consumed")
    ap.add_argument("--per-frame", action="store_true",
                    help="drive the frame loop from the host, one DragPose.run (two launches) per frame, as the reference does (default: the "
                         "frame loop runs on the device, DragPose.run_frames; same results)")
    ap.add_argument("--lockstep", action="store_true",
                    help="directory input: advance all files together, one kernel launch per frame index for all of them "
                         "(same per-file results; the reference evaluates them one after the other)")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("eval_drag needs a ROCm GPU: dragposer_amd has no CPU fallback")
    cfg = dict(DEFAULT_CONFIG)
    if args.config is not None:
        with open(args.config) as f:
            cfg = json.load(f)
    files = [args.input_path]
    if os.path.isdir(args.input_path):
        files = sorted(os.path.join(args.input_path, f) for f in os.listdir(args.input_path) if f.endswith(".bvh"))
        if not files:
            raise SystemExit(f"{args.input_path}: no .bvh files")
    raw = load_model_arrays(args.model_path, skeleton_bvh=files[0])  # (a model folder carries no skeleton: the evaluated BVH's, train.py:329-341)
    opt = LatentOptimizer(device=args.device, arrays=raw)
    encoder = PoseEncoder(arrays=raw).to(opt.device)
    tck = args.temporal_checkpoint
    if tck is None and os.path.isdir(args.model_path) and os.path.exists(os.path.join(args.model_path, "temporal.pt")):
        tck = os.path.join(args.model_path, "temporal.pt")  # train_temporal.load_model (train_temporal.py:474-482)
    if tck is not None:
        temporal_pack = load_reference_checkpoint(tck, opt.device)
    else:
        temporal_pack = (None, np.zeros(24, np.float32), np.ones(24, np.float32))
    if args.lockstep and len(files) > 1:
        print(f"Evaluate {len(files)} files in lock-step ------------------------")
        return evaluate_files(args, files, opt, encoder, temporal_pack, cfg, raw)
    results = []
    for f in files:
        print(f"Evaluate {f} ------------------------")
        results.append(evaluate_file(args, f, opt, encoder, temporal_pack, cfg, raw))
    return results


if __name__ == "__main__":
    main()
