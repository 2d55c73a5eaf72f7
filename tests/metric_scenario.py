"""Seeded synthetic evaluation scenario shared by tests/golden/make_golden.py (which feeds it to the reference's
EvalMetricsTracker) and the metric tests: 3 frames, one target each, a prediction and 5 samples per frame."""
import numpy as np

METRICS = ["PVE", "PVE-SC", "PVE-PA", "PVE-T", "PVE-T-SC", "MPJPE", "MPJPE-SC", "MPJPE-PA",
           "PVE_samples_min", "PVE-SC_samples_min", "PVE-PA_samples_min", "PVE-T_samples_min", "PVE-T-SC_samples_min",
           "MPJPE_samples_min", "MPJPE-SC_samples_min", "MPJPE-PA_samples_min"]


def make_frames(num_frames=3, num_samples=5, seed=0):
    rs = np.random.RandomState(seed)
    frames = []
    for _ in range(num_frames):
        tgt_v = (rs.randn(1, 6890, 3) * 0.3).astype(np.float32)
        tgt_r = (rs.randn(1, 6890, 3) * 0.3).astype(np.float32)
        tgt_j = (rs.randn(1, 14, 3) * 0.3).astype(np.float32)

        def noisy(t, n):
            # a random similarity transform of the target plus noise: exercises rotation, scale and translation
            q = rs.randn(n, 3, 3)
            r = np.stack([np.linalg.qr(q[i])[0] * np.sign(np.linalg.det(np.linalg.qr(q[i])[0])) for i in range(n)])
            sc = rs.uniform(0.8, 1.2, size=(n, 1, 1))
            return (sc * np.einsum("nij,nkj->nki", r, np.repeat(t, n, 0)) + rs.randn(n, 1, 3) * 0.1
                    + 0.02 * rs.randn(n, *t.shape[1:])).astype(np.float32)

        pred = {"verts": noisy(tgt_v, 1), "reposed_verts": noisy(tgt_r, 1), "joints3D": noisy(tgt_j, 1),
                "verts_samples": noisy(tgt_v, num_samples), "reposed_verts_samples": noisy(tgt_r, num_samples),
                "joints3D_samples": noisy(tgt_j, num_samples)}
        target = {"verts": tgt_v, "reposed_verts": tgt_r, "joints3D": tgt_j}
        frames.append((pred, target))
    return frames
