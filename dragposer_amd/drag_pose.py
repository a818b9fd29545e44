"""`DragPose`: the reference's stateful per-frame operator (python/src/drag_pose.py:12-414) for S
sequences advancing in lock-step, with the optimise loop dispatched to the HIP kernel.

What stays here (PyTorch-ROCm device tensors, a handful of tiny ops per frame):
  * the state the reference keeps between frames: latent, current global position /
    rotation, the 60-deep latent / displacement / height ring buffers, the temporal target buffer;
  * the temporal target block (row a12, drag_pose.py:234-294), which calls the Transformer;
  * nothing of the per-frame epilogue (row a11, drag_pose.py:369-402: global pose update, joint adjustment,
    buffer shifts, root channels of the returned pose): that is `dp_sequence_advance`, one launch on the state
    tensors held here.
What does not: decode / FK / loss / backward / Adam / the while-condition -- those run inside
`LatentOptimizer.optimize` (one kernel launch per frame index for all S sequences).

`run()` keeps the reference's argument names and meaning; tensors carry a leading sequence
dimension S (a single sequence may omit it, then the results omit it too, like the reference).
"""
import numpy as np
import torch

from .encoder import PoseEncoder
from .model import DEFAULT_MODEL, NJ
from .optimizer import LATENT, LatentOptimizer
from .temporal import HISTORY, PAST_FRAMES, SAMPLE_STEP


def arrays_from_generator_model(generator_model, offsets):
    """The reference's Generator_Model (generator_architecture.py: `.autoencoder`, `.data`, `.parents`) -> the array dict
    dragposer_amd.model.HostModel / PoseEncoder read (same keys as data/model_dancedb.npz).  `offsets` [22,3] is the
    skeleton's OFFSET table: the reference hands it to every run() (drag_pose.py:201), the kernel context wants it once."""
    sd = generator_model.autoencoder.state_dict()
    arrs = {k: v.detach().cpu().numpy() for k, v in sd.items()}
    d = generator_model.data
    arrs["means.dqs"], arrs["stds.dqs"] = d.mean_dqs.detach().cpu().numpy().reshape(-1), d.std_dqs.detach().cpu().numpy().reshape(-1)
    arrs["means.displacement"] = d.mean_displacement.detach().cpu().numpy().reshape(-1)
    arrs["stds.displacement"] = d.std_displacement.detach().cpu().numpy().reshape(-1)
    arrs["parents"] = np.asarray(generator_model.parents, dtype=np.int32)
    arrs["offsets"] = np.asarray(torch.as_tensor(offsets).detach().cpu(), dtype=np.float32).reshape(NJ, 3)
    return arrs


# what a frame step consumes of the kernel's results (the global rotation matrices `rot`, 198 floats per frame, are not among
# them: the kernel skips them when the pointer is NULL)
_RUN_OUTPUTS = ("z", "z_pre", "pose", "disp", "world_disp", "world_rot", "pos", "loss", "iters")


class DragPose:
    def __init__(self, generator_model, temporal_model, means_latent, stds_latent, device=None, device_gpu=None, n_sequences=1,
                 offsets=None, native_temporal=False):
        """Argument order of the reference (drag_pose.py:13).  `generator_model` is one of
          * a `LatentOptimizer` (an existing kernel context; `device` arguments are then ignored),
          * None / a path to a model .npz / a dict of its arrays (dragposer_amd.model) -- a context is created on
            `device_gpu` (or `device` if that names a GPU, else cuda:0),
          * the reference's own Generator_Model object, with `offsets` [22,3] (see arrays_from_generator_model).
        `n_sequences`: sequences advancing in lock-step (the reference: 1).  `native_temporal`: run the temporal target
        block (drag_pose.py:248-292) in one HIP launch (dp_temporal_predict) instead of PyTorch ops around nn.Transformer."""
        if isinstance(generator_model, LatentOptimizer) or hasattr(generator_model, "host_model"):
            optimizer, arrays = generator_model, None
        else:
            dev = next((d for d in (device_gpu, device) if d is not None and torch.device(d).type == "cuda"), "cuda:0")
            if generator_model is None or isinstance(generator_model, str):
                arrays = None
                optimizer = LatentOptimizer(model_path=generator_model or DEFAULT_MODEL, device=dev)
            else:
                if isinstance(generator_model, dict):
                    arrays = generator_model
                else:
                    if offsets is None:
                        raise ValueError("a reference Generator_Model carries no bone offsets: pass offsets=[22,3]")
                    arrays = arrays_from_generator_model(generator_model, offsets)
                optimizer = LatentOptimizer(device=dev, arrays=arrays)
        self.opt = optimizer
        self._arrays = arrays
        self._encoder = None
        self.device = optimizer.device
        self.temporal = temporal_model.to(self.device).eval() if temporal_model is not None else None
        self.S = int(n_sequences)
        dev = self.device
        hm = optimizer.host_model.arrays
        self.means_q = torch.from_numpy(hm["mean_q"]).to(dev)  # drag_pose.py:27-33
        self.stds_q = torch.from_numpy(hm["std_q"]).to(dev)
        self.offsets = torch.from_numpy(hm["offsets"]).to(dev)
        self.means_latent = torch.as_tensor(means_latent, dtype=torch.float32, device=dev)
        self.stds_latent = torch.as_tensor(stds_latent, dtype=torch.float32, device=dev)
        self.temporal_frames_index = list(PAST_FRAMES)
        self.target_latent_buffer = None
        self._native_temporal = None
        if native_temporal and self.temporal is not None:
            from .temporal import NativeTemporal

            self._native_temporal = NativeTemporal(self.temporal, self.means_latent, self.stds_latent, device=self.device)
        self.latent = None
        self.last = None
        self._idx_cache = {}
        self._trk_cache = {}
        self._out = [None, None]
        self._flip = 0
        self._offsets_checked = False

    def _index(self, values):
        """device index tensor for a Python list, created once (no host->device copy inside a captured step)"""
        key = tuple(int(v) for v in values)
        if key not in self._idx_cache:
            self._idx_cache[key] = torch.tensor(key, dtype=torch.int64, device=self.device)
        return self._idx_cache[key]

    # ------------------------------------------------------------------ state (drag_pose.py:47-64)
    def set_initial_pose(self, initial_pose, init_global_pos, initial_global_rot, initial_heights, eps=None, generator=None):
        """The reference's entry point (drag_pose.py:47): `initial_pose` (S,176,1) normalised dual quaternions ->
        latent = mu + eps * exp(logvar / 2) through the pose encoder (autoencoder.py:19-27,56-143), then the state of
        set_initial_state.  `eps` [S,24]: the normal draw to use (the reference takes it from torch's global generator);
        `generator`: a torch.Generator for the draw otherwise."""
        S = self.S
        if self._encoder is None:
            self._encoder = (PoseEncoder(arrays=self._arrays) if self._arrays is not None else PoseEncoder()).to(self.device)
        pose = torch.as_tensor(initial_pose, dtype=torch.float32, device=self.device).reshape(S, 176)
        with torch.no_grad():
            mu, logvar = self._encoder(pose)
            if eps is None:
                eps = torch.randn((S, LATENT), generator=generator)
            latent = mu + torch.as_tensor(eps, dtype=torch.float32).reshape(S, LATENT).to(self.device) * torch.exp(0.5 * logvar)
        self.set_initial_state(latent, init_global_pos, initial_global_rot, initial_heights)

    def set_initial_state(self, latent, init_global_pos, initial_global_rot, initial_heights):
        """What set_initial_pose leaves behind, with the initial latent given instead of encoded."""
        dev, S = self.device, self.S
        f = lambda t, shape: torch.as_tensor(t, dtype=torch.float32, device=dev).reshape(shape).clone()
        self.latent = f(latent, (S, LATENT))
        self.current_global_pos = f(init_global_pos, (S, 3))
        self.current_global_rot = f(initial_global_rot, (S, 4))
        self.latent_buffer = self.latent.unsqueeze(1).repeat(1, HISTORY, 1)
        self.displacement_buffer = torch.zeros(S, HISTORY, 3, device=dev)
        self.heights_buffer = f(initial_heights, (S, 1, -1)).repeat(1, HISTORY, 1)
        self.current_index = 0
        self.target_latent_buffer = None

    # ------------------------------------------------------------------ temporal target (drag_pose.py:234-294)
    def _temporal_targets(self, window):
        S, dev = self.S, self.device
        assert window % SAMPLE_STEP == 0
        if self.target_latent_buffer is None or self.target_latent_buffer.shape[1] != window + 1:
            if self.target_latent_buffer is not None:
                # a window CHANGED in the middle of a window restarts the window (a prediction is made this frame), as the native plug-in
                # does (dp_unity.cpp): the reference indexes the re-allocated buffer with the old index -- an IndexError when the window
                # shrank, rows of zeros until the next prediction when it grew -- and here the old index would make run_frames() ask
                # for a stretch of <= 0 frames
                self.current_index = 0
            self.target_latent_buffer = torch.zeros(S, window + 1, LATENT, device=dev)
        if self.current_index != 0 or self.temporal is None:  # no predictor: the buffer stays zero, use lambda_temporal = 0
            return
        if self._native_temporal is not None:
            self._native_temporal.predict(self.latent_buffer, self.displacement_buffer, self.heights_buffer, window,
                                          out=self.target_latent_buffer)
            return
        idx = self.temporal_frames_index
        idx_t = self._index(idx)
        with torch.no_grad():
            input_latent = self.latent_buffer.index_select(1, idx_t)[:, :-1].clone()
            input_disp = torch.stack([self.displacement_buffer[:, j:j + SAMPLE_STEP].sum(dim=1) for j in idx[:-1]], dim=1)
            tgt = self.latent_buffer[:, idx[-1]].unsqueeze(1).clone()
            input_latent = (input_latent - self.means_latent) / self.stds_latent
            tgt = (tgt - self.means_latent) / self.stds_latent
            heights = self.heights_buffer.index_select(1, idx_t)[:, :-1].clone()
            enc_in = torch.cat((input_latent, input_disp, heights), dim=-1)
            buf = self.target_latent_buffer
            for i in range(0, window + 1, SAMPLE_STEP):
                pred = self.temporal(enc_in, tgt)
                tgt = torch.cat((tgt, pred[:, -1:]), dim=1)
                buf[:, i] = pred[:, -1]
            buf = buf * self.stds_latent + self.means_latent
            for i in range(0, window, SAMPLE_STEP):  # "lerp" with weight 1: hold the next prediction
                buf[:, i:i + SAMPLE_STEP + 1] = buf[:, i + SAMPLE_STEP].unsqueeze(1)
            self.target_latent_buffer = buf

    # ------------------------------------------------------------------ one frame (drag_pose.py:196-414)
    def _trackers(self, mask_joints, weights_joints):
        """device copies of the tracker list and the dense per-joint weight / flag arrays of the C ABI, built once per
        distinct (mask_joints, weights_joints) -- they are the same objects frame after frame"""
        mj_h = np.asarray(mask_joints.cpu() if isinstance(mask_joints, torch.Tensor) else mask_joints, dtype=np.int64).reshape(-1)
        wj_h = np.asarray(weights_joints.cpu() if isinstance(weights_joints, torch.Tensor) else weights_joints, dtype=np.float32).reshape(-1, 2)
        key = (mj_h.tobytes(), wj_h.tobytes())
        if key not in self._trk_cache:
            if len(self._trk_cache) >= 8:  # a caller that keeps changing its tracker set: keep the cache bounded
                self._trk_cache.pop(next(iter(self._trk_cache)))
            dev, S = self.device, self.S
            if wj_h.shape[0] != mj_h.shape[0]:
                raise ValueError("target_ee_pos / target_ee_rot / weights_joints must have one row per entry of mask_joints")
            mj = torch.from_numpy(mj_h).to(dev)
            w = torch.zeros(S, NJ, 2, device=dev)
            tracked = torch.zeros(S, NJ, dtype=torch.uint8, device=dev)
            w.index_copy_(1, mj, torch.from_numpy(wj_h).to(dev).unsqueeze(0).expand(S, -1, -1).contiguous())
            tracked.index_fill_(1, mj, 1)
            # targets of untracked joints are never read, so the dense target arrays are allocated once and only the
            # tracked rows are rewritten every frame
            self._trk_cache[key] = dict(mj=mj, mj_host=mj_h, w=w, tracked=tracked, tgt_pos=torch.zeros(S, NJ, 3, device=dev),
                                        tgt_rot=torch.zeros(S, NJ, 9, device=dev))
        return self._trk_cache[key]

    @property
    def temporal_status(self):
        """DP_TEMPORAL_* bits of the native predictor's handle, read without synchronising (0 without one): 1 once a team of workgroups of an earlier
        prediction gave up waiting for a member -- that prediction's targets are NaN for the affected sequences (their frames then carry
        DP_STATUS_BAD_TARGETS in last_status), and the next prediction raises DragPoserError(DP_ERR_TIMEOUT) once before the operator goes on
        with one workgroup per sequence (include/dragposer.h: dp_temporal_status)"""
        return self._native_temporal.status() if self._native_temporal is not None else 0

    def run_frames(self, target_ee_pos, target_ee_rot, mask_joints, weights_joints, target_root=None, stop_eps_pos=1e-2, stop_eps_rot=1e-2,
                   max_iter=100, min_loss_incr=0.00001, learning_rate=1e-3, lambda_rot=1, lambda_temporal=1, temporal_future_window=60,
                   height_indices=(0, 4, 8, 13, 17, 21), joint_adjustment_indices=None, joint_adjustment_weight=0.01):
        """T consecutive frames of every sequence -- T calls of run() -- with the frame loop on the device: one kernel launch per
        stretch of frames between two temporal predictions (all T of them when there is no predictor or lambda_temporal is 0).
        target_ee_pos [T,S,E,3], target_ee_rot [T,S,E,3,3]; `target_root` [T,S,3] or None: given, the position targets of frame t
        are target_ee_pos[t] + (target_root[t] - current_global_pos) as eval_drag builds them (eval_drag.py:186-199), which no
        caller can do ahead of time.  Returns (poses [T,S,88], global positions [T,S,3], iterations [T,S])."""
        dev, S = self.device, self.S
        tp = torch.as_tensor(target_ee_pos, dtype=torch.float32, device=dev)
        T = int(tp.shape[0])
        tp = tp.reshape(T, S, -1, 3)
        tR = torch.as_tensor(target_ee_rot, dtype=torch.float32, device=dev).reshape(T, S, -1, 9)
        trk = self._trackers(mask_joints, weights_joints)
        E = trk["mj"].numel()
        if tp.shape[2] != E or tR.shape[2] != E:
            raise ValueError("target_ee_pos / target_ee_rot / weights_joints must have one row per entry of mask_joints")
        dense_p = torch.zeros(T, S, NJ, 3, device=dev)
        dense_r = torch.zeros(T, S, NJ, 9, device=dev)
        dense_p.index_copy_(2, trk["mj"], tp)
        dense_r.index_copy_(2, trk["mj"], tR)
        root = torch.as_tensor(target_root, dtype=torch.float32, device=dev).reshape(T, S, 3).contiguous() if target_root is not None else None
        adjust = None
        if joint_adjustment_indices is not None:
            joint_index, ee_index = joint_adjustment_indices
            adjust = (int(joint_index), int(trk["mj_host"][ee_index]), float(joint_adjustment_weight))
        poses = torch.empty(T, S, 88, device=dev)
        gpos = torch.empty(T, S, 3, device=dev)
        iters = torch.empty(T, S, dtype=torch.int32, device=dev)
        status = torch.empty(T, S, dtype=torch.int32, device=dev)
        # The reference predicts at current_index == 0 whatever lambda_temporal is (drag_pose.py:235-291), so the stretches between two
        # predictions are cut the same way with the pull term on or off; without a predictor there is nothing to pull towards and the
        # term is off in run() and here alike (the reference cannot run without one).
        have = self.temporal is not None
        pull = have and float(lambda_temporal) != 0.0
        window = int(temporal_future_window)
        zero_tgt = torch.zeros(S, LATENT, device=dev)
        t = 0
        while t < T:
            if have:  # frames up to the next prediction (a prediction every `window` frames; every frame when window = 0)
                self._temporal_targets(window)
                n = min(T - t, max(window, 1) - self.current_index)
                z_tgt, strides = self.target_latent_buffer[:, self.current_index:], (LATENT, (window + 1) * LATENT)
            else:
                n, z_tgt, strides = T - t, zero_tgt, (0, LATENT)
            self.opt.optimize_sequence(self.latent, dense_p[t:t + n], dense_r[t:t + n], root[t:t + n] if root is not None else None,
                                       trk["w"], trk["tracked"], z_tgt, strides, self.current_global_pos, self.current_global_rot,
                                       self.latent_buffer, self.displacement_buffer, self.heights_buffer, tuple(int(h) for h in height_indices),
                                       n_iter=max_iter, lr=learning_rate, lambda_rot=float(lambda_rot), lambda_tmp=float(lambda_temporal) if pull else 0.0,
                                       stop_eps_pos=stop_eps_pos, stop_eps_rot=stop_eps_rot, min_loss_incr=min_loss_incr, adjust=adjust,
                                       pose_ret=poses[t:t + n], pos_ret=gpos[t:t + n], iters=iters[t:t + n], status=status[t:t + n])
            t += n
            self.current_index = 0 if window == 0 else (self.current_index + n) % window
        self.last_status = status  # [T,S] DP_STATUS_* bits (include/dragposer.h): non-zero where a frame's inputs were not finite
        return poses, gpos, iters

    def run(self, target_ee_pos, target_ee_rot, mask_joints, weights_joints, offsets=None, stop_eps_pos=1e-2,
            stop_eps_rot=1e-2, max_iter=100, min_loss_incr=0.00001, learning_rate=1e-3, lambda_rot=1, lambda_temporal=1,
            temporal_future_window=60, height_indices=(0, 4, 8, 13, 17, 21), joint_adjustment_indices=None,
            joint_adjustment_weight=0.01, verbose=False, out_pose=None, out_pos=None):
        """One frame for every sequence: one launch for the optimise loop and the epilogue, one for the history buffers, plus two
        row scatters of the targets.  `out_pose` [S,88] / `out_pos` [S,3]: optional caller storage for the returned tensors."""
        dev, S = self.device, self.S
        squeeze = torch.as_tensor(target_ee_pos).dim() == 2
        tp = torch.as_tensor(target_ee_pos, dtype=torch.float32, device=dev).reshape(S, -1, 3)
        tR = torch.as_tensor(target_ee_rot, dtype=torch.float32, device=dev).reshape(S, -1, 9)
        trk = self._trackers(mask_joints, weights_joints)
        E = trk["mj"].numel()
        if tp.shape[1] != E or tR.shape[1] != E:
            raise ValueError("target_ee_pos / target_ee_rot / weights_joints must have one row per entry of mask_joints")
        if offsets is not None and not self._offsets_checked:
            if not torch.allclose(torch.as_tensor(offsets, dtype=torch.float32, device=dev).reshape(NJ, 3), self.offsets, atol=1e-6):
                raise ValueError("offsets differ from the skeleton the optimiser context was created with")
            self._offsets_checked = True  # (a device->host round trip: once, not per frame)

        self._temporal_targets(temporal_future_window)
        trk["tgt_pos"].index_copy_(1, trk["mj"], tp)
        trk["tgt_rot"].index_copy_(1, trk["mj"], tR)

        # one frame = a whole-sequence launch of one step (dp_optimize_sequence): the optimise loop AND run()'s epilogue
        # (drag_pose.py:369-402) in one kernel, the same one run_frames() uses for stretches of frames -- so cutting a sequence into
        # calls of run() or handing it to run_frames() whole gives the same bits
        adjust = None
        if joint_adjustment_indices is not None:
            joint_index, ee_index = joint_adjustment_indices
            adjust = (int(joint_index), int(trk["mj_host"][ee_index]), float(joint_adjustment_weight))
        pose = out_pose if out_pose is not None else torch.empty(S, 88, device=dev)
        gpos = out_pos if out_pos is not None else torch.empty(S, 3, device=dev)
        self._flip ^= 1  # two result sets, alternated: `last` stays valid while the next frame runs
        if self._out[self._flip] is None:
            self._out[self._flip] = dict(iters=torch.empty(1, S, dtype=torch.int32, device=dev), loss=torch.empty(1, S, 3, device=dev),
                                         status=torch.empty(1, S, dtype=torch.int32, device=dev),
                                         scratch=torch.empty(1, S, LATENT + 3 + len(height_indices), device=dev), z=torch.empty(S, LATENT, device=dev))
        o = self._out[self._flip]
        z_tgt = self.target_latent_buffer[:, self.current_index:]
        pull = self.temporal is not None and float(lambda_temporal) != 0.0  # (no predictor: the buffer is zeros, the term is off; run_frames() alike)
        self.opt.optimize_sequence(self.latent, trk["tgt_pos"].unsqueeze(0), trk["tgt_rot"].unsqueeze(0), None, trk["w"], trk["tracked"], z_tgt,
                                   (0, (int(temporal_future_window) + 1) * LATENT), self.current_global_pos, self.current_global_rot,
                                   self.latent_buffer, self.displacement_buffer, self.heights_buffer, tuple(int(h) for h in height_indices),
                                   n_iter=max_iter, lr=learning_rate, lambda_rot=float(lambda_rot), lambda_tmp=float(lambda_temporal) if pull else 0.0,
                                   stop_eps_pos=stop_eps_pos, stop_eps_rot=stop_eps_rot, min_loss_incr=min_loss_incr, adjust=adjust,
                                   pose_ret=pose.unsqueeze(0), pos_ret=gpos.unsqueeze(0), iters=o["iters"], loss=o["loss"], scratch=o["scratch"],
                                   status=o["status"])
        o["z"].copy_(self.latent)  # (self.latent is advanced in place by the next frame; `last` must not move with it)
        self.last = dict(iters=o["iters"][0], loss=o["loss"][0], z=o["z"], pose=pose, pos=gpos, status=o["status"][0])
        if verbose:
            l, it = self.last["loss"].cpu(), self.last["iters"].cpu()
            print(f"Loss sqrt(Pos): {l[:, 0].sqrt().mean():.5f} // Loss Rot: {l[:, 1].mean():.5f} // "
                  f"Loss Temporal: {l[:, 2].mean():.5f} // Iter: {it.float().mean():.1f}")
        self.current_index = 0 if temporal_future_window == 0 else (self.current_index + 1) % temporal_future_window
        if squeeze and S == 1:
            return pose[0], gpos[0]
        return pose, gpos
