"""Minimal BVH reader / writer for the evaluation pipeline (the reference uses pymotion.io.bvh.BVH,
train.py:322-326,476,508).  Hierarchy text is kept verbatim so that a result file differs from its
source only in the MOTION block."""
import numpy as np


class BVH:
    def __init__(self):
        self.names, self.parents, self.offsets, self.channels = [], [], [], []
        self.header_lines = []
        self.frame_time = 1.0 / 120.0
        self.motion = None  # [F, n_channels]

    # ------------------------------------------------------------------
    def load(self, path):
        stack, pending, in_end = [], None, False
        with open(path) as f:
            lines = f.read().splitlines()
        i = 0
        while i < len(lines):
            tok = lines[i].split()
            if tok and tok[0] == "MOTION":
                break
            self.header_lines.append(lines[i])
            if tok:
                if tok[0] in ("ROOT", "JOINT"):
                    pending = len(self.names)
                    self.names.append(tok[1])
                    self.parents.append(stack[-1] if stack else -1)
                    self.offsets.append([0.0, 0.0, 0.0])
                    self.channels.append([])
                elif tok[0] == "End":
                    in_end, pending = True, None
                elif tok[0] == "{":
                    stack.append(pending if pending is not None else -2)
                elif tok[0] == "}":
                    if stack.pop() == -2:
                        in_end = False
                elif tok[0] == "OFFSET" and not in_end:
                    self.offsets[stack[-1]] = [float(t) for t in tok[1:4]]
                elif tok[0] == "CHANNELS":
                    self.channels[stack[-1]] = tok[2:2 + int(tok[1])]
            i += 1
        n_frames = int(lines[i + 1].split()[1])
        self.frame_time = float(lines[i + 2].split()[2])
        self.motion = np.array([[float(t) for t in l.split()] for l in lines[i + 3:i + 3 + n_frames]], dtype=np.float64)
        return self

    # ------------------------------------------------------------------
    @property
    def n_joints(self):
        return len(self.names)

    def rot_order(self):
        return ["".join(c[0].lower() for c in ch if c.endswith("rotation")) for ch in self.channels]

    def get_data(self):
        """-> euler rotations [F, J, 3] (degrees, channel order), positions [F, J, 3] (root: channel values,
        others: offsets), parents (root -> 0, train.py:338), offsets (root zeroed, train.py:340), rot orders."""
        F, J = self.motion.shape[0], self.n_joints
        rot = np.zeros((F, J, 3))
        pos = np.tile(np.asarray(self.offsets)[None], (F, 1, 1))
        col = 0
        for j, ch in enumerate(self.channels):
            r = 0
            for c in ch:
                if c.endswith("position"):
                    pos[:, j, "XYZ".index(c[0])] = self.motion[:, col]
                else:
                    rot[:, j, r] = self.motion[:, col]
                    r += 1
                col += 1
        parents = np.array([max(p, 0) for p in self.parents], dtype=np.int32)
        offsets = np.asarray(self.offsets, dtype=np.float64).copy()
        offsets[0] = 0.0
        return rot, pos, parents, offsets, self.rot_order()

    def set_data(self, rot_deg, root_pos):
        """euler rotations [F, J, 3] (degrees, channel order) and root positions [F, 3] -> MOTION block"""
        F = rot_deg.shape[0]
        m = np.zeros((F, sum(len(c) for c in self.channels)))
        col = 0
        for j, ch in enumerate(self.channels):
            r = 0
            for c in ch:
                if c.endswith("position"):
                    m[:, col] = root_pos[:, "XYZ".index(c[0])] if j == 0 else self.offsets[j]["XYZ".index(c[0])]
                else:
                    m[:, col] = rot_deg[:, j, r]
                    r += 1
                col += 1
        self.motion = m

    def save(self, path):
        with open(path, "w") as f:
            f.write("\n".join(self.header_lines) + "\n")
            f.write(f"MOTION\nFrames: {self.motion.shape[0]}\nFrame Time: {self.frame_time:.6f}\n")
            for row in self.motion:
                f.write(" ".join(f"{v:.6f}" for v in row) + "\n")
