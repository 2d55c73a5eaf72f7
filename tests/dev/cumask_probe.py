"""Dev probe: does hipExtStreamCreateWithCUMask partition the chip here, and how do mask bits map to XCDs?
Times the fused mesh kernel (MFMA-bound, fills every CU) on streams restricted to subsets of the 256 CUs."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hierarchicalprobabilistic3dhuman_amd import smpl_data  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL  # noqa: E402

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def main():
    dev = torch.device("cuda:0")
    smpl = SMPL(smpl_data.synthetic_smpl_model(0)).to(dev)
    M = 6528
    g = torch.Generator().manual_seed(0)
    R = torch.linalg.qr(torch.randn(M, 24, 3, 3, generator=g))[0].to(dev)
    betas = torch.randn(M, 10, generator=g).to(dev)
    cases = [("all (default stream)", None), ("bits 0-255", range(256)), ("bits 0-127", range(128)), ("bits 0-63", range(64)),
             ("even bits", range(0, 256, 2)), ("bits = 0 mod 8 (32 CUs)", range(0, 256, 8)),
             ("bits 0-31", range(32)), ("bits 0-159", range(160)), ("bits 160-255", range(160, 256))]
    for name, bits in cases:
        st = torch.cuda.current_stream() if bits is None else masked_stream(bits)
        with torch.cuda.stream(st):
            smpl.lbs_events = []
            for _ in range(6):
                smpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=betas, pose2rot=False)
            st.synchronize()
            ms = sorted(e0.elapsed_time(e1) for (_, e0, e1) in smpl.lbs_events[2:])
        print("%-28s mesh kernel %.3f ms (min %.3f)" % (name, ms[len(ms) // 2], ms[0]), flush=True)
    # two streams with complementary masks running the same kernel at once: each should take its own masked time
    a, b = masked_stream(range(160)), masked_stream(range(160, 256))
    ev = {}
    for name, st in (("160", a), ("96", b)):
        with torch.cuda.stream(st):
            smpl.lbs_events = []
            for _ in range(4):
                smpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=betas, pose2rot=False)
            ev[name] = list(smpl.lbs_events)
    torch.cuda.synchronize()
    for name, evs in ev.items():
        print("concurrent, %s CUs: %s" % (name, ["%.3f" % e0.elapsed_time(e1) for (_, e0, e1) in evs]))


if __name__ == "__main__":
    main()
