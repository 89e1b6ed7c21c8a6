import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_ops_gpu as T
hip.load()
for S, causal in [(128, False), (96, False), (77, True), (581, False)]:
    for split in [(3, 3)]:
        Bn, Hh, hd = 1, 1, 64
        D = Hh * hd
        qkv = T.rnd(Bn * S, 3 * D, seed=17)
        Q = T.dev_h2(hip, qkv)
        out = hip.H2.empty(Bn * S, D); out.t.fill_(float("nan"))
        hip.attention(Q, out, Bn, S, Hh, hd, mode=0, causal=causal, split_qk=split[0], split_pv=split[1])
        x = Q.float().cpu().double().reshape(Bn, S, 3, Hh, hd).permute(2, 0, 3, 1, 4)
        ref = T.ref_attention(x[0], x[1], x[2], hd ** -0.5, causal=causal).permute(0, 2, 1, 3).reshape(Bn * S, D)
        got = out.float().cpu().double()
        err = (got - ref).abs()
        bad = (err > 2e-6 * ref.abs().max()).nonzero()
        print(f"S={S} causal={causal} relerr={float(err.max()/ref.abs().max()):.3e} nbad={len(bad)} of {err.numel()}")
        for r, c in bad[:24].tolist():
            hi, lo = float(out.hi[r, c]), float(out.lo[r, c])
            rh = float(ref[r, c].float().half()); rl = float((ref[r, c] - rh))
            print(f"   row {r} col {c}: ref {float(ref[r,c]):+.7f} got {float(got[r,c]):+.7f} hi {hi:+.7f} lo {lo:+.3e} | ref hi {rh:+.7f} ref lo {rl:+.3e}")
