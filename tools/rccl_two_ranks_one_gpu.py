#!/usr/bin/env python3
"""Can RCCL run the tile gather with TWO ranks on ONE device?  (The builder's box has one GPU; the 2-GPU test in tests/test_tiles_gpu.py
skips there.)  Starts the two rank processes of that test with both contexts on device 0 and compares their gathered images with the
one-rank image, bit for bit.  RCCL may refuse duplicate devices at ncclCommInitRank: then this prints the error and exits 3.
python tools/rccl_two_ranks_one_gpu.py"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def main():
    import test_tiles_gpu as t
    src = (t._RANK_SCRIPT % dict(root=ROOT)).replace("ctx = capi.Context(rank)", "ctx = capi.Context(0)")
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "rank.py")
        open(script, "w").write(src)
        procs = [subprocess.Popen([sys.executable, script, str(r), "2", tmp], stderr=subprocess.PIPE, text=True) for r in range(2)]
        rcs, errs = [], []
        for p in procs:
            try:
                _, err = p.communicate(timeout=60)
            except subprocess.TimeoutExpired:
                p.kill()
                _, err = p.communicate()
                err = "TIMEOUT after 60 s\n" + (err or "")
            rcs.append(p.returncode)
            errs.append(err or "")
        if rcs != [0, 0]:
            print("ranks exited with", rcs)
            for e in errs:
                print(e[-800:])
            return 3
        ref = np.load(os.path.join(tmp, "ref.npy"))
        for r in range(2):
            got = np.load(os.path.join(tmp, "rank%d.npy" % r))
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "rank %d differs from the one-rank image" % r
        print("two RCCL ranks on one device: both gathered images equal the one-rank image bit for bit (%dx%d)" % (ref.shape[1], ref.shape[0]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
