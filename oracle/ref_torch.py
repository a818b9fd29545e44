"""ORACLE (test infrastructure, not product): PyTorch-CPU restatement of DragPoser's hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (dragposer_amd/) never does: it fails loudly when the HIP library is missing.

What is restated (reference file:line it follows, all under /root/reference/python/src):
  decoder_forward     autoencoder.py:224-256  (f_latent, 3x [unpool, masked conv k=1, LeakyReLU 0.2
                      except last]; denorm, per-joint quaternion normalise, re-normalise)
                      skeleton.py:117-130 (SkeletonConv = (W*mask) x + b), skeleton.py:244-245 (unpool)
  quat_to_rotmat      utils.py:34-76      (to_matrix_4, w-first Hamilton, 3x3 block)
  fk_world            utils.py:80-106 + 109-149 (root-space -> parent-local rot-mats, then the
                      joint-by-joint chain T_j = T_parent T_j)
  frame_losses        drag_pose.py:66-127,185-194 (denorm, world root rotation, world displacement,
                      FK, weighted position / rotation MSE over the tracked joints, temporal MSE)
  optimize            drag_pose.py:296-355 (decode -> loss -> backward -> Adam, N times; outputs are
                      those of the LAST forward pass, z_final is after the last step), Adam =
                      torch.optim.Adam defaults re-created per frame (drag_pose.py:218)
Third-party arithmetic not under /root/reference: upc-pymotion==0.1.10 quat_torch.{mul, mul_vec,
inverse, normalize} (call sites drag_pose.py:88,102; utils.py:96; autoencoder.py:248) restated
as the standard w-first Hamilton forms; **parity unpinned at that boundary only** (the reference
holds no tests/vectors for it).  Everything else is pinned by tests/golden/*.npz, which were
produced by running the real reference (tools/make_goldens.py) and by the SURVEY 8.1 anchors.

Everything is batched over frames (leading dim B); per-frame tracked sets are expressed by
dense per-joint arrays: w[B,22,2] (0 where untracked), tracked[B,22] (bool).
"""
import json
import math
import os

import numpy as np
import torch

NJ = 22
LATENT = 24
HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_MODEL = os.path.join(os.path.dirname(HERE), "dragposer_amd", "data", "model_dancedb.npz")


class OracleModel:
    """Decoder tensors + dataset statistics + skeleton, as torch CPU tensors."""

    def __init__(self, path=DEFAULT_MODEL, dtype=torch.float32, weight_rounding="none"):
        raw = np.load(path)
        t = lambda k: torch.tensor(raw[k], dtype=torch.float32)
        self.dtype = dtype
        self.Wf, self.bf = t("decoder.f_latent.weight"), t("decoder.f_latent.bias")
        self.U, self.W, self.b = [], [], []
        for l in range(3):
            self.U.append(t(f"decoder.layers.{l}.0.weight"))
            w = t(f"decoder.layers.{l}.1.weight")[..., 0]
            m = t(f"decoder.layers.{l}.1.mask")[..., 0]
            if weight_rounding == "bf16":
                w = w.to(torch.bfloat16).float()
            self.W.append(w * m)  # skeleton.py:120
            self.b.append(t(f"decoder.layers.{l}.1.bias"))
        if weight_rounding == "bf16":
            self.Wf = self.Wf.to(torch.bfloat16).float()
        mean_dqs, std_dqs = t("means.dqs"), t("stds.dqs")
        # first 4 of every 8 dual-quaternion channels (drag_pose.py:27-33, autoencoder.py:242-244)
        self.mu4 = mean_dqs.reshape(NJ, 8)[:, :4].reshape(88).clone()
        self.sd4 = std_dqs.reshape(NJ, 8)[:, :4].reshape(88).clone()
        self.mu_d, self.sd_d = t("means.displacement"), t("stds.displacement")
        self.parents = [int(p) for p in raw["parents"]]
        self.offsets = t("offsets")
        for k in ("Wf", "bf", "mu4", "sd4", "mu_d", "sd_d", "offsets"):
            setattr(self, k, getattr(self, k).to(dtype))
        self.U = [u.to(dtype) for u in self.U]
        self.W = [w.to(dtype) for w in self.W]
        self.b = [b.to(dtype) for b in self.b]


# ---------------------------------------------------------------- quaternions (pymotion boundary)
def quat_mul(a, b):
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack(
        (
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ),
        dim=-1,
    )


def quat_rotate(q, v):
    qv = q[..., 1:]
    t = 2.0 * torch.linalg.cross(qv, v, dim=-1)
    return v + q[..., :1] * t + torch.linalg.cross(qv, t, dim=-1)


def quat_conj(q):
    return torch.cat((q[..., :1], -q[..., 1:]), dim=-1)


def quat_to_rotmat(q):
    """3x3 block of the reference's to_matrix_4 (utils.py:49-74); no normalisation."""
    w, x, y, z = q.unbind(-1)
    x2, y2, z2 = x + x, y + y, z + z
    xx, yy, zz = x * x2, y * y2, z * z2
    xy, xz, yz = x * y2, x * z2, y * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    rows = (
        torch.stack((1.0 - (yy + zz), xy - wz, xz + wy), dim=-1),
        torch.stack((xy + wz, 1.0 - (xx + zz), yz - wx), dim=-1),
        torch.stack((xz - wy, yz + wx, 1.0 - (xx + yy)), dim=-1),
    )
    return torch.stack(rows, dim=-2)


# ---------------------------------------------------------------- decoder
def decoder_raw(model, z):
    """The 7 dense products exactly as the reference executes them -> y[B,92] (pre-denorm)."""
    h = z @ model.Wf.T + model.bf
    for l in range(3):
        h = h @ model.U[l].T
        h = h @ model.W[l].T + model.b[l]
        if l != 2:
            h = torch.nn.functional.leaky_relu(h, 0.2)
    return h


def decoder_forward(model, z):
    """-> motion[B,88] (normalised-space unit quaternions), disp[B,3] (normalised space)."""
    y = decoder_raw(model, z)
    motion, disp = y[:, :88], y[:, 88:91]
    q = (motion * model.sd4 + model.mu4).reshape(-1, NJ, 4)
    q = q / torch.linalg.norm(q, dim=-1, keepdim=True)
    motion = (q.reshape(-1, 88) - model.mu4) / model.sd4
    return motion, disp


# ---------------------------------------------------------------- kinematics
def fk_world(model, q_root_space, root_pos):
    """q_root_space[B,22,4] with joint 0 already the WORLD root rotation; root_pos[B,3].

    Follows the reference's two steps: parent-local matrices (R_j for children of the root,
    R_parent^-1 R_j otherwise, utils.py:95-105), then the sequential chain (utils.py:140-146).
    Returns pos[B,22,3], rot[B,22,3,3] (global).
    """
    R = quat_to_rotmat(q_root_space)
    Rinv = quat_to_rotmat(quat_conj(q_root_space))
    par = model.parents
    G = [R[:, 0]]
    P = [root_pos]
    for j in range(1, NJ):
        p = par[j]
        local = R[:, j] if p == 0 else Rinv[:, p] @ R[:, j]
        G.append(G[p] @ local)
        P.append((G[p] @ model.offsets[j].reshape(3, 1)).squeeze(-1) + P[p])
    return torch.stack(P, dim=1), torch.stack(G, dim=1)


def pose_fk(model, motion, disp, cur_rot):
    """motion/disp in normalised space -> world_disp, world_rot, pos, rot (drag_pose.py:84-113)."""
    q = (motion * model.sd4 + model.mu4).reshape(-1, NJ, 4)
    d = disp * model.sd_d + model.mu_d
    world_rot = quat_mul(cur_rot, q[:, 0])
    q = torch.cat((world_rot.unsqueeze(1), q[:, 1:]), dim=1)
    world_disp = quat_rotate(world_rot, d)
    pos, rot = fk_world(model, q, world_disp)
    return world_disp, world_rot, pos, rot, d


def frame_losses(model, z, motion, disp, cur_rot, z_tgt, tgt_pos, tgt_rot, w, tracked, lam_rot, lam_tmp):
    """Per-frame (loss_pos, loss_rot*lam_rot, loss_tmp*lam_tmp), each [B]; plus FK outputs."""
    world_disp, world_rot, pos, rot, d = pose_fk(model, motion, disp, cur_rot)
    trk = tracked.to(z.dtype)
    E = trk.sum(dim=1)
    dp = ((pos - tgt_pos) ** 2).sum(-1) * w[..., 0] * trk
    dr = ((rot.reshape(-1, NJ, 9) - tgt_rot) ** 2).sum(-1) * w[..., 1] * trk
    loss_pos = dp.sum(1) / (3.0 * E)  # .mean() over (E,3)      drag_pose.py:116-119
    loss_rot = lam_rot * dr.sum(1) / (9.0 * E)  # .mean() over (E,3,3)  drag_pose.py:121-124,187
    loss_tmp = lam_tmp * ((z - z_tgt) ** 2).mean(dim=1)  # drag_pose.py:127,188
    return loss_pos, loss_rot, loss_tmp, dict(world_disp=world_disp, world_rot=world_rot, pos=pos, rot=rot, disp=d)


# ---------------------------------------------------------------- optimiser
def adam_scalars(n_iter, lr=1e-2, betas=(0.9, 0.999)):
    """Per-step scalars torch's single-tensor Adam computes in Python doubles."""
    out = []
    for t in range(1, n_iter + 1):
        bc1 = 1.0 - betas[0] ** t
        bc2 = 1.0 - betas[1] ** t
        out.append((lr / bc1, math.sqrt(bc2)))
    return out


def optimize(model, z0, z_tgt, cur_rot, tgt_pos, tgt_rot, w, tracked, n_iter, lr=1e-2, lam_rot=1.0,
             lam_tmp=0.02, betas=(0.9, 0.999), eps=1e-8, stop_eps_pos=0.0, stop_eps_rot=0.0,
             min_loss_incr=None, use_torch_adam=False):
    """Batched restatement of the optimise loop.  Early stopping (per frame) reproduces the
    reference's while-condition (drag_pose.py:300-304,351-355); pass stop_eps_*=0 and
    min_loss_incr=None for the fixed-iteration benchmark configuration.
    """
    dt = model.dtype
    conv = lambda a: torch.as_tensor(a).to(dt)
    z0, z_tgt, cur_rot, tgt_pos, tgt_rot, w = map(conv, (z0, z_tgt, cur_rot, tgt_pos, tgt_rot, w))
    tracked = torch.as_tensor(tracked).bool()
    B = z0.shape[0]
    z = z0.clone().requires_grad_()
    m = torch.zeros_like(z0)
    v = torch.zeros_like(z0)
    opt = torch.optim.Adam([z], lr=lr, betas=betas, eps=eps) if use_torch_adam else None
    active = torch.ones(B, dtype=torch.bool)
    prev_loss = torch.full((B,), 10000000.0, dtype=dt)
    iters = torch.zeros(B, dtype=torch.int32)
    keep = {}
    loss_hist = torch.full((B, n_iter, 3), float("nan"), dtype=dt)
    scal = adam_scalars(n_iter, lr, betas)
    min_incr = -float("inf") if min_loss_incr is None else min_loss_incr

    def latch(name, new, mask):
        if name not in keep:
            keep[name] = new.detach().clone()
        else:
            keep[name][mask] = new.detach()[mask]

    for it in range(n_iter):
        if not bool(active.any()):
            break
        motion, disp = decoder_forward(model, z)
        lp, lr_, lt, fk = frame_losses(model, z, motion, disp, cur_rot, z_tgt, tgt_pos, tgt_rot, w, tracked,
                                       lam_rot, lam_tmp)
        total = lp + lr_ + lt
        if z.grad is not None:
            z.grad = None
        total.sum().backward()
        g = z.grad.detach()
        for name, val in (("z_pre", z), ("pose", motion), ("disp_norm", disp), ("world_disp", fk["world_disp"]),
                          ("world_rot", fk["world_rot"]), ("pos", fk["pos"]), ("rot", fk["rot"]),
                          ("disp", fk["disp"]), ("grad", g)):
            latch(name, val, active)
        loss_hist[active, it] = torch.stack((lp, lr_, lt), dim=1).detach()[active]
        if use_torch_adam:
            assert bool(active.all()), "torch.optim.Adam path is for fixed-iteration runs"
            opt.step()
        else:
            step_size, bc2_sqrt = scal[it]
            with torch.no_grad():
                a = active.unsqueeze(1)
                m_new = m + (1.0 - betas[0]) * (g - m)  # exp_avg.lerp_(grad, 1-beta1)
                v_new = v * betas[1] + (1.0 - betas[1]) * g * g  # mul_(beta2).addcmul_(g, g, 1-beta2)
                denom = v_new.sqrt() / bc2_sqrt + eps
                z_new = z - step_size * (m_new / denom)  # addcdiv_(exp_avg, denom, value=-step_size)
                m = torch.where(a, m_new, m)
                v = torch.where(a, v_new, v)
                z.copy_(torch.where(a, z_new, z))
        iters += active.to(torch.int32)
        with torch.no_grad():
            tot = total.detach()
            incr = prev_loss - tot
            prev_loss = torch.where(active, tot, prev_loss)
            cont = ((lp.detach() > stop_eps_pos) | (lr_.detach() > stop_eps_rot)) & (incr > min_incr)
            active = active & cont
    out = {k: v_.numpy() for k, v_ in keep.items()}
    out["z_final"] = z.detach().numpy()
    out["iters"] = iters.numpy()
    out["loss_hist"] = loss_hist.numpy()
    last = np.clip(out["iters"] - 1, 0, None)
    out["loss"] = out["loss_hist"][np.arange(B), last]
    return out


def optimize_reference_shaped(model, z0, z_tgt, cur_rot, tgt_pos, tgt_rot, w, tracked, n_iter, **kw):
    """B=1 sequential loop with autograd + torch.optim.Adam, as the reference runs (CPU baseline)."""
    outs = []
    for b in range(len(z0)):
        s = slice(b, b + 1)
        outs.append(optimize(model, z0[s], z_tgt[s], cur_rot[s], tgt_pos[s], tgt_rot[s], w[s], tracked[s],
                             n_iter, use_torch_adam=True, **kw))
    return {k: np.concatenate([o[k] for o in outs], axis=0) for k in outs[0]}


# ---------------------------------------------------------------- synthetic recipe S
TRACK6 = [0, 3, 7, 13, 17, 21]
TRACK3 = [13, 17, 21]
W6 = {0: (10.0, 10.0), 3: (5.0, 0.01), 7: (5.0, 0.01), 13: (5.0, 0.01), 17: (5.0, 0.01), 21: (5.0, 0.01)}
W3 = {13: (20.0, 20.0), 17: (5.0, 0.01), 21: (5.0, 0.01)}


def synth_inputs(model, B, trackers=6, mixed=False, seed=1234):
    """Recipe S: draws (Zs, Z0, ZT, CR[, Eb, perms]) and targets = FK(decode(Zs), CR)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    Zs = torch.randn(B, 24, generator=g) * 0.3
    Z0 = torch.randn(B, 24, generator=g) * 0.3
    ZT = Z0 + 0.05 * torch.randn(B, 24, generator=g)
    CR = torch.randn(B, 4, generator=g)
    CR = CR / torch.linalg.norm(CR, dim=-1, keepdim=True)
    wtab = W6 if trackers == 6 else W3
    w = torch.zeros(B, NJ, 2)
    tracked = torch.zeros(B, NJ, dtype=torch.bool)
    if mixed:
        Eb = torch.randint(1, 7, (B,), generator=g)
        for b in range(B):
            perm = torch.randperm(6, generator=g)
            for i in perm[: int(Eb[b])]:
                j = TRACK6[int(i)]
                tracked[b, j] = True
                w[b, j] = torch.tensor(W6[j])
    else:
        for j, wj in wtab.items():
            tracked[:, j] = True
            w[:, j] = torch.tensor(wj)
    with torch.no_grad():
        dt = model.dtype
        motion, disp = decoder_forward(model, Zs.to(dt))
        _, _, pos, rot, _ = pose_fk(model, motion, disp, CR.to(dt))
    trk = tracked.unsqueeze(-1).to(dt)
    return dict(
        z_src=Zs.numpy(), z0=Z0.numpy(), z_tgt=ZT.numpy(), cur_rot=CR.numpy(),
        tgt_pos=(pos * trk).float().numpy(), tgt_rot=(rot.reshape(B, NJ, 9) * trk).float().numpy(),
        w=w.numpy(), tracked=tracked.numpy().astype(np.uint8),
    )


def load_golden(path):
    raw = np.load(path)
    out = {k: raw[k] for k in raw.files if k != "meta"}
    out["meta"] = json.loads(bytes(raw["meta"]).decode())
    return out
