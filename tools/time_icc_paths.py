"""ICC refine on the bench's scene: us per iteration of the one-launch path (k_icc_iter) and of the two-launch path
(MF_ICC_TWO_LAUNCH=1), 1 and 8 scenes per GPU, and the bit-identity of what the two paths return."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Workload, parse  # noqa: E402

out = {}
for scenes in (1, 8):
    sys.argv = [sys.argv[0], "--scenes-per-gpu", str(scenes)]
    args = parse()
    wl = Workload(args, 0, torch.device("cuda", 0))
    res = {}
    for two in (0, 1):
        os.environ["MF_ICC_ONE_LAUNCH"] = str(1 - two)

        def run(n):
            wl.q.copy_(wl.q0); wl.t.copy_(wl.t0); wl.m.zero_(); wl.v.zero_()
            wl.icc.refine(wl.q, wl.t, wl.m, wl.v, n, step0=0, alpha_q=0.01, alpha_t=0.001)

        def timed(n, reps=10):
            for _ in range(3):
                run(n)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                run(n)
            e.record(); torch.cuda.synchronize()
            return s.elapsed_time(e) / reps

        t100, t20 = timed(100), timed(20)
        run(100)
        torch.cuda.synchronize()
        res[two] = dict(ms_100=t100, us_iter=(t100 - t20) / 80 * 1e3, us_iter_incl_fixed=t100 / 100 * 1e3,
                        q=wl.q.clone(), t=wl.t.clone(), m=wl.m.clone(), v=wl.v.clone())
    same = all(torch.equal(res[0][k], res[1][k]) for k in "qtmv")
    out[f"scenes{scenes}"] = dict(one_launch_us_iter=round(res[0]["us_iter"], 3), two_launch_us_iter=round(res[1]["us_iter"], 3),
                                  one_launch_ms_100=round(res[0]["ms_100"], 4), two_launch_ms_100=round(res[1]["ms_100"], 4),
                                  bit_identical=same)
    del wl
print(json.dumps(out))
