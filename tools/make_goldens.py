#!/usr/bin/env python3
"""Generate the golden vectors and the model fixture from the REAL reference code.

Runs only in the build container (needs /root/reference).  Nothing here is imported by the
product, the tests or the bench; the outputs are data files:

  dragposer_amd/data/model_dancedb.npz   decoder/encoder tensors + dataset statistics + skeleton
                                         (weights are CC BY-SA 4.0, see dragposer_amd/data/NOTICE)
  tests/golden/{s1,s3,s4,es}.npz         inputs and expected outputs of DragPose.run()
  tests/golden/enc.npz                   the real Encoder and DragPose.set_initial_pose on 24 poses

How the reference is driven (all reference code is executed from /root/reference, nothing is
copied): the reference modules are imported with tools/pymotion_standin on sys.path (the
un-vendored dependency upc-pymotion==0.1.10 is absent; the stand-in restates the four torch
quaternion helpers the hot path calls).  For every synthetic frame the *real*
``DragPose.run()`` (drag_pose.py:196-414) is executed: real Decoder, real loss()/FK, real
autograd and real torch.optim.Adam.  The only substitutions are harness-side:
  * the temporal predictor (weights missing from the mount: temporal.pt) is replaced by a stub
    that returns the recipe's z_tgt, with means_latent=0 / stds_latent=1, so that
    ``target_latent`` (drag_pose.py:294) is exactly the stored input z_tgt;
  * ``set_initial_pose`` (encoder + randn) is bypassed: z0 / cur_rot are stored inputs.

Synthetic recipe S: seed 1234, draw order Zs, Z0, ZT-noise, CR
(+ Eb, perms for S4); targets = FK(decode(Zs), CR) at the tracked joints.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/python"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REPO, "tools", "pymotion_standin"))
sys.path.insert(0, os.path.join(REF, "src"))

torch.set_num_threads(1)

import drag_pose as ref_drag_pose  # noqa: E402  (reference)
import train as ref_train  # noqa: E402  (reference)
import utils as ref_utils  # noqa: E402  (reference)
from generator_architecture import Generator_Model  # noqa: E402  (reference)
from train_data import Train_Data  # noqa: E402  (reference)

NJ = 22
TRACK6 = [0, 3, 7, 13, 17, 21]
TRACK3 = [13, 17, 21]


def parse_bvh_skeleton(path):
    """Own minimal BVH HIERARCHY reader: joint OFFSET rows (End Sites skipped) and parents."""
    parents, offsets, stack = [], [], []
    in_end_site = False
    pending = None
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "MOTION":
                break
            if tok[0] in ("ROOT", "JOINT"):
                pending = len(parents)
                parents.append(stack[-1] if stack else 0)
                offsets.append(None)
            elif tok[0] == "End":
                in_end_site = True
                pending = None
            elif tok[0] == "{":
                stack.append(pending if pending is not None else -1)
            elif tok[0] == "}":
                if stack.pop() == -1:
                    in_end_site = False
            elif tok[0] == "OFFSET" and not in_end_site:
                offsets[stack[-1]] = [float(t) for t in tok[1:4]]
    parents[0] = 0  # train.py:338
    offsets = np.asarray(offsets, dtype=np.float32)
    offsets[0] = 0.0  # train.py:340
    return np.asarray(parents, dtype=np.int32), offsets


class StubTemporal:
    """Stands in for the missing temporal.pt: returns the recipe's z_tgt as the prediction."""

    device = "cpu"

    def __init__(self):
        self.z_tgt = None

    def eval(self):
        return self

    def __call__(self, input_latent, input_target_latent):
        return self.z_tgt.reshape(1, 1, -1).clone()


class RecordingDragPose(ref_drag_pose.DragPose):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.rec = []

    def loss(self, *a, **k):
        out = super().loss(*a, **k)
        self.rec.append(
            dict(
                losses=[float(out[0]), float(out[1]), float(out[2])],
                world_disp=out[4].detach().clone().reshape(3),
                world_rot=out[6].detach().clone().reshape(4),
                pos=out[7].detach().clone().reshape(NJ, 3),
                rot=_FK_STASH["rot"].detach().clone().reshape(NJ, 3, 3),
            )
        )
        return out


_FK_STASH = {}


def _recording_fk(rot, global_pos, offsets, parents):
    pos, rotm = ref_utils.fk_rotmat(rot, global_pos, offsets, parents)
    _FK_STASH["pos"], _FK_STASH["rot"] = pos, rotm
    return pos, rotm


ref_drag_pose.fk_rotmat = _recording_fk


def build_reference(parents, weight_rounding=None, dtype=torch.float32):
    """dtype = torch.float64: the SAME reference code with every module, buffer and constant in double precision
    (torch's default dtype is switched before anything is constructed; the checkpoint's fp32 values are copied into
    double parameters by load_state_dict, the dataset statistics are cast)."""
    torch.set_default_dtype(dtype)
    td = Train_Data("cpu", ref_train.param, None)
    gm = Generator_Model("cpu", ref_train.param, list(int(p) for p in parents), td)
    ref_train.load_model(gm, os.path.join(REF, "models/model_dancedb/generator.pt"), td, "cpu")
    td.set_means(td.mean_dqs.to(dtype), td.mean_displacement.to(dtype))
    td.set_stds(td.std_dqs.to(dtype), td.std_displacement.to(dtype))
    assert next(gm.autoencoder.decoder.parameters()).dtype == dtype
    if weight_rounding == "bf16":
        # S4: every decoder *weight* tensor rounded to bf16 (biases, masks, unpool untouched)
        with torch.no_grad():
            dec = gm.autoencoder.decoder
            dec.f_latent.weight.copy_(dec.f_latent.weight.to(torch.bfloat16).float())
            for layer in dec.layers:
                layer[1].weight.copy_(layer[1].weight.to(torch.bfloat16).float())
    stub = StubTemporal()
    drag = RecordingDragPose(gm, stub, torch.zeros(24), torch.ones(24), "cpu", "cpu")
    return gm, td, drag, stub


def reset_state(drag, z0, cur_rot):
    """What set_initial_pose (drag_pose.py:47-64) leaves behind, with z0/cur_rot given."""
    drag.current_global_pos = torch.zeros(1, 3, 1)
    drag.current_global_rot = cur_rot.reshape(1, 4).clone()
    drag.latent = z0.reshape(1, 24).clone().detach().requires_grad_()
    drag.latent_buffer = torch.tile(drag.latent.detach(), (60, 1))
    drag.displacement_buffer = torch.zeros(60, 3)
    drag.heights_buffer = torch.zeros(60, 6)
    drag.current_index = 0
    drag.target_latent_buffer = None
    drag.rec = []


def forward_fk(drag, td, z, cur_rot, offsets):
    """decode(z) -> FK via the reference's own loss() path; returns pos(22,3), rot(22,3,3)."""
    reset_state(drag, z, cur_rot)
    with torch.no_grad():
        motion, disp = drag.decoder(drag.latent, td.mean_dqs, td.std_dqs)
        drag.loss(
            motion, disp,
            torch.zeros(1, 1, 1, 3), torch.zeros(1, 1, 1, 3, 3), torch.zeros(24),
            offsets, torch.tensor([0]), torch.ones(1, 2), 1.0, 0.0,
        )
    r = drag.rec[-1]
    return r["pos"], r["rot"], motion.detach().reshape(88), disp.detach().reshape(3)


def draw_recipe(B, mixed):
    g = torch.Generator(device="cpu").manual_seed(1234)
    Zs = torch.randn(B, 24, generator=g) * 0.3
    Z0 = torch.randn(B, 24, generator=g) * 0.3
    ZT = Z0 + 0.05 * torch.randn(B, 24, generator=g)
    CR = torch.randn(B, 4, generator=g)
    CR = CR / torch.linalg.norm(CR, dim=-1, keepdim=True)
    tracked = None
    if mixed:
        Eb = torch.randint(1, 7, (B,), generator=g)
        tracked = []
        for b in range(B):
            perm = torch.randperm(6, generator=g)
            tracked.append(sorted(TRACK6[int(i)] for i in perm[: int(Eb[b])]))
    return Zs, Z0, ZT, CR, tracked


def run_recipe(name, B, track, cfg_weights, lam_tmp, n_iter, offsets_t, parents,
               mixed=False, weight_rounding=None, early_stop=False):
    gm, td, drag, stub = build_reference(parents, weight_rounding)
    Zs, Z0, ZT, CR, tracked = draw_recipe(B, mixed)
    out = dict(
        z0=Z0.numpy(), z_tgt=ZT.numpy(), cur_rot=CR.numpy(), z_src=Zs.numpy(),
        w=np.zeros((B, NJ, 2), np.float32), tracked=np.zeros((B, NJ), np.uint8),
        tgt_pos=np.zeros((B, NJ, 3), np.float32), tgt_rot=np.zeros((B, NJ, 9), np.float32),
        z_final=np.zeros((B, 24), np.float32), z_pre=np.zeros((B, 24), np.float32),
        pose=np.zeros((B, 88), np.float32), pose_ret=np.zeros((B, 88), np.float32),
        disp_norm=np.zeros((B, 3), np.float32),
        world_disp=np.zeros((B, 3), np.float32), world_rot=np.zeros((B, 4), np.float32),
        pos=np.zeros((B, NJ, 3), np.float32), rot=np.zeros((B, NJ, 9), np.float32),
        global_pos_ret=np.zeros((B, 3), np.float32),
        loss_hist=np.full((B, n_iter, 3), np.nan, np.float32), iters=np.zeros((B,), np.int32),
    )
    for b in range(B):
        tr = tracked[b] if mixed else track
        idx = torch.tensor(tr, dtype=torch.int64)
        wj = cfg_weights[idx]
        pos_t, rot_t, _, _ = forward_fk(drag, td, Zs[b], CR[b], offsets_t)
        tp, tR = pos_t[idx].clone(), rot_t[idx].clone()
        out["w"][b, tr] = wj.numpy()
        out["tracked"][b, tr] = 1
        out["tgt_pos"][b, tr] = tp.numpy()
        out["tgt_rot"][b, tr] = tR.reshape(-1, 9).numpy()

        reset_state(drag, Z0[b], CR[b])
        stub.z_tgt = ZT[b]
        captured = {}
        orig_decoder_forward = drag.decoder.forward

        def rec_forward(*a, **k):
            m, d = orig_decoder_forward(*a, **k)
            captured["motion"], captured["disp"] = m.detach().clone(), d.detach().clone()
            return m, d

        drag.decoder.forward = rec_forward
        if early_stop:  # the reference's own eval settings, eval_drag.py:210-214
            kw = dict(stop_eps_pos=0.01 * 0.01, stop_eps_rot=0.01, max_iter=n_iter, min_loss_incr=0.00001)
        else:
            kw = dict(stop_eps_pos=0.0, stop_eps_rot=0.0, max_iter=n_iter, min_loss_incr=-float("inf"))
        pose_ret, gpos_ret = drag.run(
            target_ee_pos=tp, target_ee_rot=tR, mask_joints=idx, weights_joints=wj,
            offsets=offsets_t, learning_rate=1e-2, lambda_rot=1, lambda_temporal=lam_tmp,
            temporal_future_window=0, height_indices=[0, 4, 8, 13, 17, 21],
            joint_adjustment_indices=None, joint_adjustment_weight=0.0, verbose=False, **kw
        )
        drag.decoder.forward = orig_decoder_forward
        it = len(drag.rec)
        last = drag.rec[-1]
        out["iters"][b] = it
        out["z_final"][b] = drag.latent.detach().reshape(24).numpy()
        out["z_pre"][b] = drag.current_latent.reshape(24).numpy()
        out["pose"][b] = captured["motion"].reshape(88).numpy()
        out["disp_norm"][b] = captured["disp"].reshape(3).numpy()
        out["pose_ret"][b] = pose_ret.detach().reshape(88).numpy()
        out["global_pos_ret"][b] = gpos_ret.detach().reshape(3).numpy()
        out["world_disp"][b] = last["world_disp"].numpy()
        out["world_rot"][b] = last["world_rot"].numpy()
        out["pos"][b] = last["pos"].numpy()
        out["rot"][b] = last["rot"].reshape(NJ, 9).numpy()
        for i, r in enumerate(drag.rec):
            out["loss_hist"][b, i] = r["losses"]
        if b % 8 == 0:
            print(f"[{name}] frame {b}/{B} iters={it} losses={last['losses']}", flush=True)
    meta = dict(name=name, B=B, n_iter=n_iter, lr=1e-2, lambda_rot=1.0, lambda_tmp=lam_tmp,
                betas=[0.9, 0.999], eps=1e-8, mixed=mixed, weight_rounding=weight_rounding or "none",
                early_stop=bool(early_stop),
                stop_eps_pos=1e-4 if early_stop else 0.0, stop_eps_rot=1e-2 if early_stop else 0.0,
                min_loss_incr=1e-5 if early_stop else None, torch=torch.__version__)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    return out


def run_frames(inp, n_iter, lam_tmp, offsets, parents, dtype=torch.float32, weight_rounding=None, progress=None):
    """The REAL DragPose.run on given inputs (dense per-joint arrays of SURVEY 8a: z0, z_tgt, cur_rot, tgt_pos, tgt_rot, w,
    tracked), fixed iteration count, in precision `dtype`; returns what the parity tests compare (fp32 arrays)."""
    gm, td, drag, stub = build_reference(parents, weight_rounding, dtype)
    offsets_t = torch.tensor(offsets, dtype=dtype)
    B = len(inp["z0"])
    out = dict(pos=np.zeros((B, NJ, 3), np.float32), z_final=np.zeros((B, 24), np.float32), z_pre=np.zeros((B, 24), np.float32),
               loss_last=np.zeros((B, 3), np.float64))
    for b in range(B):
        tr = np.nonzero(inp["tracked"][b])[0]
        idx = torch.tensor(tr, dtype=torch.int64)
        tp = torch.tensor(inp["tgt_pos"][b][tr], dtype=dtype)
        tR = torch.tensor(inp["tgt_rot"][b][tr].reshape(-1, 3, 3), dtype=dtype)
        wj = torch.tensor(inp["w"][b][tr], dtype=dtype)
        reset_state(drag, torch.tensor(inp["z0"][b], dtype=dtype), torch.tensor(inp["cur_rot"][b], dtype=dtype))
        stub.z_tgt = torch.tensor(inp["z_tgt"][b], dtype=dtype)
        drag.run(target_ee_pos=tp, target_ee_rot=tR, mask_joints=idx, weights_joints=wj, offsets=offsets_t,
                 stop_eps_pos=0.0, stop_eps_rot=0.0, max_iter=n_iter, min_loss_incr=-float("inf"), learning_rate=1e-2,
                 lambda_rot=1, lambda_temporal=lam_tmp, temporal_future_window=0, height_indices=[0, 4, 8, 13, 17, 21],
                 joint_adjustment_indices=None, joint_adjustment_weight=0.0, verbose=False)
        assert len(drag.rec) == n_iter
        last = drag.rec[-1]
        out["pos"][b] = last["pos"].numpy()
        out["z_final"][b] = drag.latent.detach().reshape(24).numpy()
        out["z_pre"][b] = drag.current_latent.reshape(24).numpy()
        out["loss_last"][b] = last["losses"]
        if progress and b % 64 == 0:
            print(f"[{progress}] frame {b}/{B}", flush=True)
    return out


def _run_frames_worker(job):
    torch.set_num_threads(1)
    return run_frames(*job[0], **job[1])


def run_frames_parallel(inp, n_iter, lam_tmp, offsets, parents, workers, **kw):
    """run_frames over `workers` forked processes (frames are independent; each worker builds its own reference objects)"""
    import multiprocessing as mp

    B = len(inp["z0"])
    cuts = np.linspace(0, B, workers + 1).astype(int)
    jobs = []
    for k in range(workers):
        sl = slice(cuts[k], cuts[k + 1])
        kwk = dict(kw)
        kwk["progress"] = (kw.get("progress") or "run") + f" w{k}" if k == 0 else None
        jobs.append((({key: v[sl] for key, v in inp.items()}, n_iter, lam_tmp, offsets, parents), kwk))
    with mp.get_context("fork").Pool(workers) as pool:
        parts = pool.map(_run_frames_worker, jobs)
    return {key: np.concatenate([p[key] for p in parts], 0) for key in parts[0]}


def inputs_digest(inp):
    import hashlib

    h = hashlib.sha256()
    for key in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked"):
        h.update(np.ascontiguousarray(inp[key]).tobytes())
    return h.hexdigest()


def add_f64(name, gold, offsets, parents, workers):
    """Re-run a committed fixture's stored inputs through the real reference in float64 and add pos_f64 / z_final_f64 /
    z_pre_f64 / loss_last_f64 to it (the fp32 arrays already in the file are left as they are) -- the parity tests define a
    "sensitive" frame as one where the REFERENCE's fp32 and fp64 runs part ways, not through the repo's own oracles."""
    path = os.path.join(gold, f"{name}.npz")
    raw = dict(np.load(path))
    meta = json.loads(bytes(raw["meta"]).decode())
    inp = {k: raw[k] for k in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")}
    o = run_frames_parallel(inp, meta["n_iter"], meta["lambda_tmp"], offsets, parents, workers, dtype=torch.float64,
                            weight_rounding=None if meta["weight_rounding"] == "none" else meta["weight_rounding"], progress=name + "_f64")
    for k in ("pos", "z_final", "z_pre", "loss_last"):
        raw[k + "_f64"] = o[k]
    np.savez_compressed(path, **raw)
    d = np.linalg.norm(raw["pos"] - raw["pos_f64"], axis=-1).max(1) * 1000
    print(f"{name}: reference fp32 vs fp64: max {d.max():.4f} mm, frames above 0.02 mm: {np.nonzero(d > 0.02)[0].tolist()} "
          f"({np.round(d[d > 0.02], 3).tolist()})")


def full_size_reference(name, B, seed, gold, offsets, parents, workers, trackers=6, mixed=False, n_iter=50, lam_tmp=0.02,
                        weight_rounding=None, keep_f64=False):
    """A BASELINE batch at size (recipe S of SURVEY 8d, B frames, inputs exactly as tests / bench.py draw them through
    oracle.ref_torch.synth_inputs) through the REAL reference in fp32 and in fp64.  The fixture holds the tracked joints' targets
    (the tests re-draw the rest and check the digest of everything), the fp32 reference positions / latents of every frame, per frame the distance between
    the reference's own fp32 and fp64 runs, and the fp64 positions of the frames where that distance exceeds 0.02 mm (keep_f64: of
    every frame -- the 1024-frame fixtures of configs 4 and 5, whose tolerance is stated as mean / p99 / max over all frames).
    trackers = 3 / n_iter = 100 / lam_tmp = 0.15: recipe S3 (config 4); mixed + weight_rounding = "bf16": recipe S4 (config 5), the
    reference's decoder weight tensors rounded to bf16 in both runs, targets drawn from that same rounded model."""
    sys.path.insert(0, REPO)
    from oracle import ref_torch as R  # inputs only: the recipe the GPU tests and bench.py use

    inp = R.synth_inputs(R.OracleModel(weight_rounding=weight_rounding or "none"), B, trackers=trackers, mixed=mixed, seed=seed)
    o32 = run_frames_parallel(inp, n_iter, lam_tmp, offsets, parents, workers, dtype=torch.float32, weight_rounding=weight_rounding,
                              progress=name + "_f32")
    o64 = run_frames_parallel(inp, n_iter, lam_tmp, offsets, parents, workers, dtype=torch.float64, weight_rounding=weight_rounding,
                              progress=name + "_f64")
    d = (np.linalg.norm(o32["pos"] - o64["pos"], axis=-1).max(1) * 1000).astype(np.float32)
    sens = np.nonzero(d > 0.02)[0].astype(np.int32)
    meta = dict(name=name, B=B, seed=seed, n_iter=n_iter, lambda_tmp=lam_tmp, trackers=trackers, mixed=bool(mixed),
                weight_rounding=weight_rounding or "none", digest=inputs_digest(inp), torch=torch.__version__)
    path = os.path.join(gold, f"{name}.npz")
    T6 = [0, 3, 7, 13, 17, 21]  # the targets travel with the fixture: they come out of CPU matrix products whose last bits differ between hosts
    extra = {}
    if mixed:
        extra["tracked6"] = inp["tracked"][:, T6]
    if keep_f64:
        extra.update(pos_f64=o64["pos"], z_pre_f64=o64["z_pre"], loss_last_f64=o64["loss_last"].astype(np.float32))
    np.savez_compressed(path, pos=o32["pos"], z_final=o32["z_final"], z_pre=o32["z_pre"], loss_last=o32["loss_last"].astype(np.float32),
                        tgt_pos6=inp["tgt_pos"][:, T6], tgt_rot6=inp["tgt_rot"][:, T6],
                        ref32_vs_ref64_mm=d, sens_frames=sens, pos_f64_sens=o64["pos"][sens], z_final_f64_sens=o64["z_final"][sens],
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **extra)
    dj = np.linalg.norm(o32["pos"] - o64["pos"], axis=-1).reshape(-1) * 1000
    print(f"wrote {path} {os.path.getsize(path)} bytes; reference fp32 vs fp64 over B x 22: mean {dj.mean():.5f} p99 {np.percentile(dj, 99):.4f} "
          f"max {dj.max():.4f} mm; > 0.02 mm on {len(sens)} frames {sens.tolist()[:40]} ({np.round(d[sens], 3).tolist()[:40]} mm)")


def export_model(parents, offsets):
    sd = torch.load(os.path.join(REF, "models/model_dancedb/generator.pt"), map_location="cpu")["model_state_dict"]
    data = torch.load(os.path.join(REF, "models/model_dancedb/data.pt"), map_location="cpu")
    arrs = {k.replace("autoencoder.", ""): v.numpy() for k, v in sd.items()}
    arrs["means.dqs"] = data["means"]["dqs"].numpy()
    arrs["means.displacement"] = data["means"]["displacement"].numpy()
    arrs["stds.dqs"] = data["stds"]["dqs"].numpy()
    arrs["stds.displacement"] = data["stds"]["displacement"].numpy()
    arrs["parents"] = parents
    arrs["offsets"] = offsets
    path = os.path.join(REPO, "dragposer_amd", "data", "model_dancedb.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes")


def encoder_golden(parents):
    """The real Encoder (autoencoder.py:56-143) and the real DragPose.set_initial_pose (drag_pose.py:47-64) on 24 poses:
    16 drawn around the dataset mean (normalised space) and 8 far from it.  set_initial_pose draws eps with the global
    torch RNG (reparameterize, autoencoder.py:19-27): it is seeded per pose and recovered from (latent - mu) / std, so the
    build can be handed the same eps."""
    gm, td, drag, stub = build_reference(parents)
    g = torch.Generator().manual_seed(4321)
    poses = torch.cat([torch.randn(16, 176, generator=g), 3.0 * torch.randn(8, 176, generator=g)], 0)
    mus, logvars, latents, epss, bufs = [], [], [], [], []
    heights = torch.tensor([0.9, 0.1, 0.1, 1.6, 0.8, 0.8])
    with torch.no_grad():
        for k in range(len(poses)):
            x = poses[k].reshape(1, 176, 1)
            mu, logvar = gm.autoencoder.encoder(x)
            torch.manual_seed(1000 + k)
            drag.set_initial_pose(x, torch.zeros(1, 3, 1), torch.tensor([1.0, 0, 0, 0]).reshape(1, 4, 1), heights)
            lat = drag.latent.detach().reshape(24)
            mus.append(mu.reshape(24)); logvars.append(logvar.reshape(24)); latents.append(lat)
            epss.append((lat - mu.reshape(24)) / torch.exp(0.5 * logvar.reshape(24)))
            assert drag.latent_buffer.shape == (60, 24) and drag.heights_buffer.shape == (60, 6)
            bufs.append(drag.latent_buffer[0].detach().clone())
    out = dict(pose=poses.numpy(), mu=torch.stack(mus).numpy(), logvar=torch.stack(logvars).numpy(), latent=torch.stack(latents).numpy(),
               eps=torch.stack(epss).numpy(), latent_buffer_row=torch.stack(bufs).numpy(), heights=heights.numpy(),
               meta=json.dumps(dict(seed=4321, eps_seeds="1000 + k", history=60)))
    return out


def anchors(parents, offsets_t):
    """Known-answer anchors A1/A2 of SURVEY.md section 8.1 -> tests/golden/anchors.npz."""
    gm, td, drag, stub = build_reference(parents)
    pos, rot, motion, disp = forward_fk(drag, td, torch.zeros(24), torch.tensor([1.0, 0, 0, 0]), offsets_t)
    a1 = dict(motion=motion.numpy(), disp_norm=disp.numpy(), pos=pos.numpy(), rot=rot.reshape(NJ, 9).numpy(),
              world_disp=drag.rec[-1]["world_disp"].numpy())
    # A2: z=0, all targets 0 / I, z_tgt = 0.1, weights of the 6-tracker config, one Adam step
    reset_state(drag, torch.zeros(24), torch.tensor([1.0, 0, 0, 0]))
    stub.z_tgt = torch.full((24,), 0.1)
    idx = torch.tensor(TRACK6)
    wj = torch.tensor([[10.0, 10.0]] + [[5.0, 0.01]] * 5)
    motion, disp = drag.decoder(drag.latent, td.mean_dqs, td.std_dqs)
    lo = drag.loss(motion, disp, torch.zeros(1, 1, 6, 3), torch.eye(3).expand(1, 1, 6, 3, 3), stub.z_tgt,
                   offsets_t, idx, wj, 1.0, 0.02)
    (lo[0] + lo[1] + lo[2]).backward()
    a2 = dict(losses=np.array([float(lo[0]), float(lo[1]), float(lo[2])], np.float64),
              grad=drag.latent.grad.reshape(24).numpy().copy())
    reset_state(drag, torch.zeros(24), torch.tensor([1.0, 0, 0, 0]))
    drag.run(target_ee_pos=torch.zeros(6, 3), target_ee_rot=torch.eye(3).expand(6, 3, 3).clone(),
             mask_joints=idx, weights_joints=wj, offsets=offsets_t, stop_eps_pos=0.0, stop_eps_rot=0.0,
             max_iter=1, min_loss_incr=-float("inf"), learning_rate=1e-2, lambda_rot=1, lambda_temporal=0.02,
             temporal_future_window=0, joint_adjustment_indices=None)
    a2["z_after_one_step"] = drag.latent.detach().reshape(24).numpy().copy()
    path = os.path.join(REPO, "tests", "golden", "anchors.npz")
    np.savez_compressed(path, **{f"a1_{k}": v for k, v in a1.items()}, **{f"a2_{k}": v for k, v in a2.items()})
    print("wrote", path)
    print("A1 motion[:8]", a1["motion"][:8], "\nA2 losses", a2["losses"], "\nA2 grad[:4]", a2["grad"][:4])


TEMPORAL_SMALL = dict(n_encoder_layers=1, n_decoder_layers=1, dim_feedforward=32)  # keeps the fixture small


def run_sequences(name, K, T, cfg, offsets_t, parents, seed):
    """K sequences x T frames through the REAL DragPose.run with all its state (warm-started latent, global
    position / rotation, ring buffers, temporal target block, joint adjustment, early stop) -- SURVEY rows a11-a13.
    The temporal predictor is the reference's Temporal class with seeded random weights (temporal.pt is
    missing from the mount) and a reduced width so that its state_dict fits a fixture."""
    import copy

    import train_temporal as ref_tt
    from temporal_transformer import Temporal

    torch.manual_seed(seed)
    tparam = copy.deepcopy(ref_tt.param)
    tparam.update(TEMPORAL_SMALL)
    temporal = Temporal(tparam, "cpu")
    temporal.eval()
    means_latent = 0.1 * torch.randn(24)
    stds_latent = 0.5 + torch.rand(24)
    td = Train_Data("cpu", ref_train.param, None)
    gm = Generator_Model("cpu", ref_train.param, list(int(p) for p in parents), td)
    ref_train.load_model(gm, os.path.join(REF, "models/model_dancedb/generator.pt"), td, "cpu")
    drag = RecordingDragPose(gm, temporal, means_latent, stds_latent, "cpu", "cpu")
    mask = torch.tensor(cfg["mask"])
    idx = torch.nonzero(mask).squeeze()
    wj = torch.tensor(cfg["weights"], dtype=torch.float32)[idx]
    E = int(idx.numel())
    height_idx = [0, 4, 8, 13, 17, 21]
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = dict(
        z0=np.zeros((K, 24), np.float32), init_rot=np.zeros((K, 4), np.float32), init_heights=np.zeros((K, 6), np.float32),
        tgt_pos=np.zeros((T, K, E, 3), np.float32), tgt_rot=np.zeros((T, K, E, 3, 3), np.float32),
        pose_ret=np.zeros((T, K, 88), np.float32), gpos_ret=np.zeros((T, K, 3), np.float32), iters=np.zeros((T, K), np.int32),
        latent=np.zeros((T, K, 24), np.float32), cur_rot=np.zeros((T, K, 4), np.float32), z_tgt=np.zeros((T, K, 24), np.float32),
        mask_idx=idx.numpy().astype(np.int32), weights=wj.numpy(), means_latent=means_latent.numpy(), stds_latent=stds_latent.numpy(),
    )
    for k in range(K):
        zgt = torch.randn(24, generator=g) * 0.3
        cr = torch.randn(4, generator=g)
        cr = cr / torch.linalg.norm(cr)
        z0 = zgt + 0.05 * torch.randn(24, generator=g)
        pos0, _, _, _ = forward_fk(drag, td, z0, cr, offsets_t)
        reset_state(drag, z0, cr)
        heights0 = pos0[height_idx, 1].clone()
        if cfg.get("unity_initial_heights"):  # what the Unity path passes (run_drag.py:93: torch.zeros(6))
            heights0 = torch.zeros(6)
        drag.heights_buffer[:] = heights0
        out["z0"][k], out["init_rot"][k], out["init_heights"][k] = z0.numpy(), cr.numpy(), heights0.numpy()
        gt_rot, gt_pos = cr.clone(), torch.zeros(3)
        for t in range(T):
            zgt = zgt + 0.02 * torch.randn(24, generator=g)
            with torch.no_grad():
                motion, disp = drag.decoder(zgt.reshape(1, 24), td.mean_dqs, td.std_dqs)
                qs = (motion * drag.stds_dqs + drag.means_dqs)[0, :, 0].reshape(NJ, 4)
                d = (disp * drag.stds_displacement + drag.means_displacement)[0, :, 0]
                gt_rot = ref_drag_pose.quat.mul(gt_rot.reshape(1, 4), qs[0:1]).reshape(4)
                gt_pos = gt_pos + ref_drag_pose.quat.mul_vec(gt_rot.reshape(1, 4), d.reshape(1, 3)).reshape(3)
                qs_t = qs.clone()
                qs_t[0] = gt_rot
                rotm = ref_utils.from_root_quat_to_rotmat(qs_t.reshape(1, 1, NJ, 4), torch.tensor(drag.parents))
                rel = (gt_pos - drag.current_global_pos.detach()[0, :, 0]).reshape(1, 1, 3)  # eval_drag.py:186
                pos_t, rot_t = ref_utils.fk_rotmat(rotm, rel, offsets_t, drag.parents)
            tp, tR = pos_t[0, 0, idx].clone(), rot_t[0, 0, idx].clone()
            n0 = len(drag.rec)
            ci = drag.current_index
            ja = tuple(cfg["joint_adjustment_indices"]) if cfg["enable_joint_adjustment"] else None
            # "lambda_switch_frame": the pull term is off before that frame and on from it (Unity's SetLambdas in the middle of a window)
            lam_t = 0.0 if t < cfg.get("lambda_switch_frame", 0) else cfg["lambda_temporal"]
            pose_ret, gpos_ret = drag.run(
                target_ee_pos=tp, target_ee_rot=tR, mask_joints=idx, weights_joints=wj, offsets=offsets_t,
                stop_eps_pos=0.01 * 0.01, stop_eps_rot=0.01, max_iter=100, min_loss_incr=0.00001, learning_rate=1e-2,
                lambda_rot=1, lambda_temporal=lam_t, temporal_future_window=cfg["temporal_future_window"],
                height_indices=height_idx, joint_adjustment_indices=ja, joint_adjustment_weight=cfg["joint_adjustment_weight"],
                verbose=False)
            out["tgt_pos"][t, k], out["tgt_rot"][t, k] = tp.numpy(), tR.numpy()
            out["pose_ret"][t, k] = pose_ret.detach().reshape(88).numpy()
            out["gpos_ret"][t, k] = gpos_ret.detach().reshape(3).numpy()
            out["iters"][t, k] = len(drag.rec) - n0
            out["latent"][t, k] = drag.latent.detach().reshape(24).numpy()
            out["cur_rot"][t, k] = drag.current_global_rot.detach().reshape(4).numpy()
            out["z_tgt"][t, k] = drag.target_latent_buffer[ci].detach().reshape(24).numpy()
        print(f"[{name}] sequence {k}: iters {out['iters'][:, k].tolist()}", flush=True)
        out[f"final_latent_buffer_{k}"] = drag.latent_buffer.detach().numpy().copy()
        out[f"final_displacement_buffer_{k}"] = drag.displacement_buffer.detach().numpy().copy()
        out[f"final_heights_buffer_{k}"] = drag.heights_buffer.detach().numpy().copy()
    for key, v in temporal.state_dict().items():
        out["temporal." + key] = v.numpy()
    meta = dict(name=name, K=K, T=T, cfg=cfg, temporal_param=TEMPORAL_SMALL, seed=seed, torch=torch.__version__)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="model,anchors,s1,s3,s4,es,seq6,seq3,sequ,enc")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--workers", type=int, default=6)
    args = ap.parse_args()
    todo = args.only.split(",")
    parents, offsets = parse_bvh_skeleton(os.path.join(REF, "data/example/eval/example.bvh"))
    assert list(parents) == [0, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 12, 11, 14, 15, 16, 11, 18, 19, 20]
    offsets_t = torch.tensor(offsets)
    gold = os.path.join(REPO, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    with open(os.path.join(REF, "config/6_trackers_config.json")) as f:
        cfg6 = json.load(f)
    with open(os.path.join(REF, "config/3_trackers_config.json")) as f:
        cfg3 = json.load(f)
    with open(os.path.join(REF, "config/4_trackers_config.json")) as f:  # the reference's third shipped configuration: pelvis + head + hands
        cfg4 = json.load(f)
    w6, w3 = torch.tensor(cfg6["weights"], dtype=torch.float32), torch.tensor(cfg3["weights"], dtype=torch.float32)
    B = args.frames
    if "model" in todo:
        export_model(parents, offsets)
    if "anchors" in todo:
        anchors(parents, offsets_t)
    jobs = {
        "s1": dict(track=TRACK6, cfg_weights=w6, lam_tmp=cfg6["lambda_temporal"], n_iter=50),
        "s3": dict(track=TRACK3, cfg_weights=w3, lam_tmp=cfg3["lambda_temporal"], n_iter=100),
        "s4": dict(track=TRACK6, cfg_weights=w6, lam_tmp=cfg6["lambda_temporal"], n_iter=50, mixed=True,
                   weight_rounding="bf16"),
        "es": dict(track=TRACK6, cfg_weights=w6, lam_tmp=cfg6["lambda_temporal"], n_iter=100, early_stop=True),
    }
    for name, kw in jobs.items():
        if name in todo:
            out = run_recipe(name, B, offsets_t=offsets_t, parents=parents, **kw)
            path = os.path.join(gold, f"{name}.npz")
            np.savez_compressed(path, **out)
            print("wrote", path, os.path.getsize(path), "bytes", flush=True)
    for name in ("s1", "s3", "s4"):
        if name + "_f64" in todo:  # adds the reference's own float64 run to a committed fixture
            add_f64(name, gold, offsets, parents, args.workers)
    # BASELINE config 2 / the headline batch / config 3's 8192-frame batch (round 5: whole, no longer a probe), through the reference
    for name, Bf in (("full1024", 1024), ("full4096", 4096), ("full8192", 8192)):
        if name in todo:
            full_size_reference(name, Bf, 1234, gold, offsets, parents, args.workers)
    if "full_s3_1024" in todo:  # BASELINE config 4 at size: 3 trackers, 100 iterations, lambda_temporal of 3_trackers_config.json
        full_size_reference("full_s3_1024", 1024, 1234, gold, offsets, parents, args.workers, trackers=3, n_iter=100,
                            lam_tmp=cfg3["lambda_temporal"], keep_f64=True)
    if "full_s4_1024" in todo:  # BASELINE config 5 at size: 1-6 trackers per frame, bf16-rounded decoder weight tensors
        full_size_reference("full_s4_1024", 1024, 1234, gold, offsets, parents, args.workers, mixed=True, n_iter=50,
                            lam_tmp=cfg6["lambda_temporal"], weight_rounding="bf16", keep_f64=True)
    if "enc" in todo:
        path = os.path.join(gold, "enc.npz")
        np.savez_compressed(path, **encoder_golden(parents))
        print("wrote", path, os.path.getsize(path), "bytes", flush=True)
    # "sequ": the shape of the Unity path (run_drag.py:141-157): one sequence, no joint adjustment, zero initial heights,
    # the temporal term on with a window that takes several autoregressive calls
    cfgu = dict(cfg6, enable_joint_adjustment=False, lambda_temporal=0.02, temporal_future_window=8, unity_initial_heights=True)
    # "sequ_switch": the same shape with the pull term switched on at frame 11 -- the fourth frame of the second window of 8: the
    # reference has been predicting at every window start all along (drag_pose.py:247-291 does not look at lambda_temporal)
    cfgs = dict(cfgu, lambda_switch_frame=11)
    for name, cfg, K, T, seed in (("seq6", cfg6, 4, 24, 77), ("seq3", cfg3, 2, 36, 78), ("seq4", cfg4, 2, 36, 81), ("sequ", cfgu, 1, 24, 79),
                                  ("sequ_switch", cfgs, 1, 24, 80)):
        if name in todo:
            out = run_sequences(name, K, T, cfg, offsets_t, parents, seed)
            path = os.path.join(gold, f"{name}.npz")
            np.savez_compressed(path, **out)
            print("wrote", path, os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    main()
