"""Stand-in for the un-vendored third-party package ``upc-pymotion==0.1.10`` (python/requirements.txt:2; absent, no network).

Used ONLY by tools/make_goldens.py and tools/make_f1_goldens.py in the build container, so that the reference's own Python
modules (which ``import pymotion...`` at module level) can be imported and RUN to produce golden vectors.  It is never imported
by the product, the tests or the bench.

  rotations/quat_torch.py   the four torch quaternion helpers the hot path calls (drag_pose.py:88,102; utils.py:29-30,96;
                            autoencoder.py:248), restated here
  everything else           the package's API surface the reference's evaluation pipeline touches (eval_drag.py, train.py:322-341,
                            409-509, motion_data.py:225-324, eval_metrics.py), as thin adapters over THIS REPO's own numpy
                            implementations (dragposer_amd/quat_np.py, bvh.py): from_euler / to_euler / unroll, dual quaternions,
                            to_root_dual_quat, fk, the BVH reader / writer.

What that buys, and what it does not: with the adapters in place the reference's OWN plumbing -- TestMotionData, get_info_from_bvh,
result_to_bvh, the per-frame target synthesis of eval_drag.main, eval_pos_error -- executes line by line and its outputs pin
dragposer_amd/motion.py, eval_drag.py and bvh.py (tests/golden/f1_*.npz).  The pymotion PRIMITIVES themselves (Euler composition
order, the dual part 0.5 t (x) r, the sign convention of unroll, BVH parsing) are this repo's on both sides of that comparison:
they remain the shared assumption, unpinned (the reference holds no vector for them).
"""
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _REPO not in sys.path:
    sys.path.append(_REPO)  # (dragposer_amd: the numpy implementations the adapters wrap)
