"""ms per call of the transform variants the prover uses, device-resident: plain NTT, iNTT (1/n fused), coeff_to_extended
(zero-pad + coset scale fused into the first pass), extended_to_coeff (1/n and coset^-1 fused into the last).  KS=20,22 picks sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
h2 = load_pkg(); h2.init(0)
import torch


def timed(f, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f(); f(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for ek in [int(v) for v in os.environ.get("KS", "20,22").split(",")]:
    d = h2.EvaluationDomain.new(4, ek - 2)
    assert d.extended_k == ek
    a = h2.gen_scalars_device(5, 1 << ek)
    r = {
        "ntt": timed(lambda: h2.ntt_device(a, d.extended_omega, ek)),
        "intt": timed(lambda: h2.ifft_device(a, d.extended_omega_inv, ek, d.extended_ifft_divisor)),
        "coeff_to_extended": timed(lambda: h2.coeff_to_extended_device(a, ek - 2, ek, d.extended_omega, d.g_coset, d.g_coset_inv)),
        "extended_to_coeff": timed(lambda: h2.extended_to_coeff_device(a, ek, d.extended_omega_inv, d.extended_ifft_divisor, d.g_coset, d.g_coset_inv)),
    }
    print("2^%d: " % ek + "  ".join("%s %.4f" % kv for kv in r.items()), flush=True)
