#!/bin/bash
# Where a step of a whole-sequence launch (dp_optimize_sequence, one sequence) spends its cycles: a diagnostic build of dp_w4.hip
# with -DDP_SEQ_STAMPS (s_memtime stamps accumulated over the steps, written over the first floats of `loss`), linked against the
# product build's other objects.  Run on the GPU box: tools/seq_step_stamps.sh > gpurun_out/sequence_step_stamps.txt
set -e
mkdir -p _scratch
B=dragposer_amd/csrc/_build
FLAGS=$(python3 -c "import __graft_entry__ as g; print(' '.join(f for f in g.HIPCC_FLAGS if f != '-shared'))")
hipcc $FLAGS -DDP_SEQ_STAMPS -c dragposer_amd/csrc/dp_w4.hip -o _scratch/dp_w4_stamps.o
hipcc --offload-arch=gfx950 -shared -fPIC -o _scratch/lib_seqstamps.so $B/dp_host.o $B/dp_w16_host.o _scratch/dp_w4_stamps.o $B/dp_w16.o $B/dp_w16_2w.o $B/dp_w16_es.o $B/dp_w16_2w_es.o $B/dp_w16_long.o $B/dp_w16_2w_long.o $B/dp_w16_es_long.o $B/dp_w16_2w_es_long.o $B/dp_sequence.o $B/dp_temporal.o
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import ref_torch as R
from dragposer_amd.optimizer import LatentOptimizer, to_device_batch
dev = torch.device("cuda:0")
opt = LatentOptimizer(device=dev, _lib_path=os.path.join(os.getcwd(), "_scratch", "lib_seqstamps.so"))
g = R.load_golden(os.path.join("tests", "golden", "es.npz"))
T, S = 2000, 1
b = R.synth_inputs(R.OracleModel(), T)
d1 = to_device_batch({k: g[k][:1] for k in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")}, dev)
tp = torch.from_numpy(b["tgt_pos"]).to(dev).reshape(T, 1, 22, 3).contiguous(); tr = torch.from_numpy(b["tgt_rot"]).to(dev).reshape(T, 1, 22, 9).contiguous()
for n_iter in (1, 8):
    lat = d1["z0"].clone(); gp, gr = torch.zeros(S, 3, device=dev), d1["cur_rot"].clone()
    lb, db, hb = torch.zeros(S, 60, 24, device=dev), torch.zeros(S, 60, 3, device=dev), torch.zeros(S, 60, 6, device=dev)
    r = opt.optimize_sequence(lat, tp, tr, None, d1["w"], d1["tracked"], torch.zeros(S, 24, device=dev), (0, 24), gp, gr, lb, db, hb, (0, 4, 8, 13, 17, 21),
                              lambda_tmp=0.0, n_iter=n_iter, stop_eps_pos=0.0, stop_eps_rot=0.0, min_loss_incr=float("-inf"))
    torch.cuda.synchronize()
    st = r["loss"].reshape(-1)[:5].cpu().numpy() / T
    print(f"{n_iter} iteration(s) per step, one sequence, {T} steps: shader cycles per step: targets + warm start {st[0]:.0f}, iterations {st[1]:.0f}, "
          f"outputs of the last forward pass {st[2]:.0f}, latent / count stores {st[3]:.0f}, run()'s state update {st[4]:.0f}; total {st.sum():.0f}")
PY
