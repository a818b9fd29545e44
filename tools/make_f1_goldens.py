#!/usr/bin/env python3
"""Pin the evaluation pipeline (BASELINE config 1) with the REFERENCE'S OWN plumbing.

Runs only in the build container (needs /root/reference).  Nothing here is imported by the product, the tests or the bench;
the outputs are data files tests/golden/f1_*.npz.

What is executed: the reference's `eval_drag.main(args)` (python/src/eval_drag.py:21-252), unmodified, from /root/reference --
its argument handling, `train.get_info_from_bvh` (train.py:329-341), `TestMotionData.add_motion / normalize`
(motion_data.py:225-324), `DragPose.set_initial_pose / run` (drag_pose.py), the per-frame target synthesis
(eval_drag.py:164-202), `train.result_to_bvh` (train.py:437-509) and `eval_metrics.eval_pos_error` (eval_metrics.py:6-32).
Harness-side only:
  * `pymotion` (absent) is tools/pymotion_standin: this repo's numpy primitives behind pymotion's API.  The primitives are
    therefore common to both sides of every comparison these fixtures support; what the fixtures pin is everything the
    reference builds ON them.
  * `temporal.pt` is missing from the mount: main() is pointed at a temporary model folder holding copies of generator.pt /
    data.pt and a seeded, random-weight temporal.pt.  Two kinds of run: `lambda_temporal = 0` in a temporary copy of the
    tracker config (the predictor still runs, as the reference does, but its output multiplies zero: results do not depend on
    the weights -- what `python -m dragposer_amd.eval_drag` does when no temporal checkpoint exists), and the config as
    shipped with a NARROW Temporal (train_temporal.param's layer counts / feed-forward width reduced in place before main()
    builds the model) whose state_dict travels in the fixture, so the test can hand the same predictor to the product.
  * recorders wrapped around the reference's classes (subclasses / function wrappers; the wrapped code runs unchanged).
"""
import argparse
import copy
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/python"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REPO, "tools", "pymotion_standin"))
sys.path.insert(0, os.path.join(REF, "src"))
torch.set_num_threads(1)

import drag_pose as ref_drag_pose  # noqa: E402  (reference)
import eval_drag as ref_eval  # noqa: E402  (reference)
import eval_metrics as ref_metrics  # noqa: E402  (reference)
import motion_data as ref_motion_data  # noqa: E402  (reference)
import train as ref_train  # noqa: E402  (reference)
import train_temporal as ref_tt  # noqa: E402  (reference)
from temporal_transformer import Temporal  # noqa: E402  (reference)

REC = {}
PERTURB = {"eps": 0.0}


class RecDragPose(ref_drag_pose.DragPose):
    """the reference's operator with its inputs / outputs written down"""

    def set_initial_pose(self, initial_pose, init_global_pos, initial_global_rot, initial_heights):
        with torch.no_grad():
            mu, logvar = self.encoder(initial_pose)
        super().set_initial_pose(initial_pose, init_global_pos, initial_global_rot, initial_heights)
        if PERTURB["eps"]:  # the reference's own closed-loop sensitivity: the same run from a latent moved by about one fp32 ulp
            with torch.no_grad():
                self.latent += PERTURB["eps"] * torch.tensor([1.0, -1.0] * 12).reshape(1, 24)
            self.latent_buffer = torch.tile(self.latent.detach(), (self.latent_buffer.shape[0], 1))
        lat = self.latent.detach().reshape(24)
        REC["initial_pose"] = initial_pose.detach().reshape(176).numpy().copy()
        REC["initial_latent"] = lat.numpy().copy()
        REC["initial_mu"], REC["initial_logvar"] = mu.reshape(24).numpy().copy(), logvar.reshape(24).numpy().copy()
        REC["initial_heights"] = torch.as_tensor(initial_heights).detach().reshape(-1).numpy().copy()
        REC["frames"] = []
        self._n_loss = 0

    def loss(self, *a, **k):
        self._n_loss += 1
        return super().loss(*a, **k)

    def run(self, **kw):
        n0 = self._n_loss
        gpos_before = self.current_global_pos.detach().reshape(3).numpy().copy()
        pose, gpos = super().run(**kw)
        REC["frames"].append(dict(tp=kw["target_ee_pos"].detach().numpy().copy(), tR=kw["target_ee_rot"].detach().numpy().copy(),
                                  pose=pose.detach().reshape(88).numpy().copy(), gpos=gpos.detach().reshape(3).numpy().copy(),
                                  iters=self._n_loss - n0, gpos_before=gpos_before, latent=self.latent.detach().reshape(24).numpy().copy()))
        REC["run_kwargs"] = {k: (v if isinstance(v, (int, float, bool, type(None))) else None) for k, v in kw.items()}
        return pose, gpos


class RecTestMotionData(ref_motion_data.TestMotionData):
    def normalize(self):
        super().normalize()
        m = self.norm_motions[0]
        REC["motion"] = {k: m[k].detach().numpy().copy() for k in ("dqs", "displacement", "global_pos", "global_rot", "heights")}


def _rec_eval_pos_error(*a, **k):
    out = _orig_eval_pos_error(*a, **k)
    REC["mpjpe"], REC["mpeepe"] = out
    return out


_orig_eval_pos_error = ref_metrics.eval_pos_error
ref_metrics.eval_pos_error = _rec_eval_pos_error
ref_eval.DragPose = RecDragPose
ref_eval.TestMotionData = RecTestMotionData

FULL_TEMPORAL = copy.deepcopy(ref_tt.param)
NARROW = dict(n_encoder_layers=1, n_decoder_layers=1, dim_feedforward=32)  # keeps the state_dict small enough for a fixture


def run_reference(bvh_path, config_name, temporal_on, seed, max_frames=None):
    """-> REC of one eval_drag.main() call"""
    REC.clear()
    ref_tt.param.clear()
    ref_tt.param.update(copy.deepcopy(FULL_TEMPORAL))
    if temporal_on:
        ref_tt.param.update(NARROW)
    work = tempfile.mkdtemp(prefix="f1_")
    cwd = os.getcwd()
    try:
        model_dir = os.path.join(work, "model")
        os.makedirs(model_dir)
        os.makedirs(os.path.join(work, "data"))
        for f in ("generator.pt", "data.pt"):
            shutil.copyfile(os.path.join(REF, "models/model_dancedb", f), os.path.join(model_dir, f))
        torch.manual_seed(seed)
        tm = Temporal(ref_tt.param, "cpu")
        means_latent, stds_latent = 0.1 * torch.randn(24), 0.5 + torch.rand(24)
        torch.save({"model_state_dict": tm.state_dict(), "means_latent": means_latent, "stds_latent": stds_latent}, os.path.join(model_dir, "temporal.pt"))
        with open(os.path.join(REF, "config", config_name)) as f:
            cfg = json.load(f)
        if not temporal_on:
            cfg["lambda_temporal"] = 0.0
        cfg_path = os.path.join(work, config_name)
        with open(cfg_path, "w") as f:
            json.dump(cfg, f)
        src = bvh_path
        if max_frames is not None:  # a clip of the first frames, cut with plain text tools
            lines = open(bvh_path).read().splitlines()
            k = next(i for i, l in enumerate(lines) if l.split() and l.split()[0] == "MOTION")
            src = os.path.join(work, os.path.basename(bvh_path))
            with open(src, "w") as f:
                f.write("\n".join(lines[:k + 1] + [f"Frames: {max_frames}", lines[k + 2]] + lines[k + 3:k + 3 + max_frames]) + "\n")
        os.chdir(work)
        ref_eval.main(argparse.Namespace(model_path=model_dir, input_path=src, config=cfg_path, verbose=False))
        out = dict(REC)
        out["cfg"] = cfg
        out["result_motion"] = np.array([[float(t) for t in l.split()] for l in
                                         open(os.path.join(work, "data", "eval_" + os.path.basename(src))).read().split("Frame Time:")[1].splitlines()[1:] if l.strip()])
        if temporal_on:
            out["temporal_sd"] = {k: v.numpy().copy() for k, v in tm.state_dict().items()}
            out["means_latent"], out["stds_latent"] = means_latent.numpy(), stds_latent.numpy()
        return out
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


def pack(name, r, stride, bvh_name, temporal_on, seed, twin=None):
    """the fixture: everything for a short clip (stride 1), a strided sample plus whole-sequence scalars for the long file"""
    fr = r["frames"]
    T = len(fr)
    idx = np.unique(np.concatenate((np.arange(0, min(T, 64)), np.arange(0, T, stride)))).astype(np.int32)
    m = r["motion"]
    out = dict(
        sample=idx, n_frames=np.int32(T),
        # TestMotionData (motion_data.py:225-324), normalised
        dqs=m["dqs"][idx].astype(np.float32), displacement=m["displacement"][idx].astype(np.float32),
        global_pos=m["global_pos"][0].T.astype(np.float32), global_rot=m["global_rot"][0].T.astype(np.float32), heights=m["heights"].astype(np.float32),
        dqs_colsum=m["dqs"].astype(np.float64).sum(0), dqs_colabs=np.abs(m["dqs"].astype(np.float64)).sum(0),
        # set_initial_pose (drag_pose.py:47-64)
        initial_pose=r["initial_pose"], initial_latent=r["initial_latent"], initial_mu=r["initial_mu"], initial_logvar=r["initial_logvar"],
        initial_heights=r["initial_heights"],
        # per-frame targets as run() received them (eval_drag.py:164-202) and the root position they are relative to
        tgt_pos=np.stack([fr[i]["tp"] for i in idx]).astype(np.float32), tgt_rot=np.stack([fr[i]["tR"] for i in idx]).astype(np.float32),
        gpos_before=np.stack([f["gpos_before"] for f in fr]).astype(np.float32),
        # results: returned pose (sample), returned global position / iteration count / stepped latent (every frame)
        pose_ret=np.stack([fr[i]["pose"] for i in idx]).astype(np.float32), gpos_ret=np.stack([f["gpos"] for f in fr]).astype(np.float32),
        iters=np.array([f["iters"] for f in fr], np.int16), latent=np.stack([fr[i]["latent"] for i in idx]).astype(np.float32),
        # the written file's MOTION block (train.py:437-509) and the metrics (eval_metrics.py:6-32)
        result_motion=r["result_motion"][idx].astype(np.float32), mpjpe=np.float64(r["mpjpe"]), mpeepe=np.float64(r["mpeepe"]),
    )
    if stride == 1:
        out["pose_ret_all"] = np.stack([f["pose"] for f in fr]).astype(np.float32)
        out["result_motion_all"] = r["result_motion"].astype(np.float64)
    if twin is not None:  # the reference run again from an initial latent moved by 1e-7 (one fp32 ulp at |z| ~ 1): how far ITS OWN loop carries that
        tf = twin["frames"]
        out["twin_iters"] = np.array([f["iters"] for f in tf], np.int16)
        out["twin_gpos_ret"] = np.stack([f["gpos"] for f in tf]).astype(np.float32)
        out["twin_pose_ret"] = np.stack([tf[i]["pose"] for i in (range(T) if stride == 1 else idx)]).astype(np.float32)
        out["twin_mpjpe"], out["twin_mpeepe"] = np.float64(twin["mpjpe"]), np.float64(twin["mpeepe"])
    if temporal_on:
        for k, v in r["temporal_sd"].items():
            out["temporal." + k] = v
        out["means_latent"], out["stds_latent"] = r["means_latent"], r["stds_latent"]
    meta = dict(name=name, bvh=bvh_name, cfg=r["cfg"], temporal_on=bool(temporal_on), temporal_param=NARROW if temporal_on else None, seed=seed,
                stride=stride, run_kwargs=r["run_kwargs"], torch=torch.__version__)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(REPO, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    it = out["iters"]
    print(f"wrote {path} {os.path.getsize(path)} bytes: {T} frames, iterations/frame mean {it.mean():.2f} (min {it.min()}, max {it.max()}), "
          f"MPJPE {r['mpjpe']:.6f} MPEEPE {r['mpeepe']:.6f}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="f1_clip6,f1_clip3,f1_clip6_t,f1_clip3_t,f1_example")
    args = ap.parse_args()
    todo = args.only.split(",")
    clip = os.path.join(REPO, "tests", "data", "example_clip.bvh")  # 240 frames of the reference's example motion (committed test data)
    full = os.path.join(REF, "data/example/eval/example.bvh")       # BASELINE config 1's file, 5052 frames
    jobs = {  # name: (file, tracker config, temporal term on?, storage stride)
        "f1_clip6": (clip, "6_trackers_config.json", False, 1),
        "f1_clip3": (clip, "3_trackers_config.json", False, 1),
        "f1_clip6_t": (clip, "6_trackers_config.json", True, 1),   # window 0: a prediction every frame
        "f1_clip3_t": (clip, "3_trackers_config.json", True, 1),   # window 16
        "f1_clip4": (clip, "4_trackers_config.json", False, 1),    # the reference's third shipped configuration (pelvis + head + hands,
        "f1_clip4_t": (clip, "4_trackers_config.json", True, 1),   #   joint adjustment [0, 0] with weight 1, lambda 0.125, window 16)
        "f1_example": (full, "6_trackers_config.json", False, 8),
    }
    for name, (path, cfgname, ton, stride) in jobs.items():
        if name in todo:
            PERTURB["eps"] = 0.0
            r = run_reference(path, cfgname, ton, seed=4242)
            PERTURB["eps"] = 1e-7
            twin = run_reference(path, cfgname, ton, seed=4242)
            PERTURB["eps"] = 0.0
            pack(name, r, stride, os.path.basename(path), ton, 4242, twin=twin)


if __name__ == "__main__":
    main()
