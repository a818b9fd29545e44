"""Stand-in for the un-vendored third-party package ``upc-pymotion==0.1.10``.

Used ONLY by tools/make_goldens.py in the build container, so that the reference's own
Python modules (which ``import pymotion...`` at module level) can be imported and run to
produce golden vectors.  It is never imported by the product, the tests or the bench.

Only the four torch quaternion helpers the hot path calls are implemented
(rotations/quat_torch.py); every other name is an empty placeholder so that the reference's
module-level imports succeed.  Their semantics are the package's published ones (w-first
Hamilton quaternions) and are the one place where parity is *unpinned*: see DESIGN.md.
"""
