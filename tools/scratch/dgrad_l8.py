"""The conv data gradient (k = 3, s = 2, GELU' operand, odd frame count: batch wrap + tail columns) on the streaming loop 8 with its
operand instance: bit-compare against loop 7 and loop 4, check against fp64, time every loop.  Shapes: the student's layers at B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
dev = "cuda"; lib = L.lib(); k, s = 3, 2
st = lambda: torch.cuda.current_stream().cuda_stream

def case(B, T, cin, cout, dt, check):
    tdt = torch.bfloat16 if dt == 1 else torch.float16
    Tout = (T - k) // s + 1
    w = (torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5).to(tdt)
    wtt = w.permute(2, 1, 0).contiguous().view(k * cin, cout)
    dy = torch.randn(B, Tout, cout, device=dev).to(tdt)
    aux = torch.randn(B, T, cin, device=dev).to(tdt)
    n_even = Tout + 1 if 2 * Tout <= T - 1 else Tout
    outs, times = {}, {}
    for path in (4, 7, 8):
        dx = torch.full((B, T, cin), float("nan"), device=dev, dtype=tdt)
        d = L.GemmDesc()
        d.X = L.operand(dy.data_ptr(), 0, cout, rpb=n_even, bstride=Tout * cout, nseg=2, tshift=(0, -1), tvalid=Tout)
        d.Y = L.operand(wtt.data_ptr(), 0, cout, nseg=2, segoff=(0, 2 * cin * cout), seg1_limit=cin)
        d.C = dx.data_ptr()
        d.M, d.N, d.Kseg, d.groups, d.splitk = B * n_even, 2 * cin, cout, 1, 1
        d.ldc, d.c_rpb, d.c_bstride = 2 * cin, n_even, T * cin
        d.ep = L.epilogue(gelu_aux=aux.data_ptr(), tail_cols=cin if 2 * n_even > T else 0, aux_factor=1)
        lib.fh_gemm_set_path(path)
        rc = lib.fh_gemm(d, dt, st()); assert rc == 0, rc
        ran = lib.fh_gemm_last_loop()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            a.record()
            for _ in range(5): lib.fh_gemm(d, dt, st())
            b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / 5)
        if 2 * n_even < T: dx[:, 2 * n_even:] = 0
        outs[path] = dx; times[path] = (best, ran)
    msg = f"B={B} T={T} {cin}->{cout} dt={dt}: " + "  ".join(f"L{p}{'' if r == p else '>' + str(r)} {t:7.3f} ms" for p, (t, r) in times.items())
    same7 = torch.equal(outs[8].view(torch.int16), outs[7].view(torch.int16)); same4 = torch.equal(outs[8].view(torch.int16), outs[4].view(torch.int16))
    msg += f" | loop 8 == loop 7: {same7}, == loop 4: {same4}, NaN left: {int(torch.isnan(outs[8].float()).sum())}"
    if not same7:
        df = (outs[8].float() - outs[7].float()).abs()
        nz = (outs[8].view(torch.int16) != outs[7].view(torch.int16))
        msg += f" [differing elements {int(nz.sum())} of {nz.numel()}, max |diff| {df.max().item():.3e}, max |value| there {outs[7].float().abs()[nz].max().item():.3e}]"
    if check:
        wd = w.double()
        ref = torch.nn.grad.conv1d_input((B, cin, T), wd, dy.double().transpose(1, 2), stride=s).transpose(1, 2) * aux.double()
        errs = {pp: (outs[pp].double() - ref).abs().max().item() / ref.abs().max().item() for pp in outs}
        rms = {pp: ((outs[pp].double() - ref) ** 2).mean().sqrt().item() for pp in outs}
        msg += ", vs fp64 max-rel " + " ".join(f"L{pp} {e:.2e}" for pp, e in errs.items()) + " rms " + " ".join(f"L{pp} {e:.3e}" for pp, e in rms.items())
        msg += f" | L7 == L4: {torch.equal(outs[7].view(torch.int16), outs[4].view(torch.int16))}"
    print(msg, flush=True)
    return same7 or not check or dt == 2 or T % 2 == 0

torch.manual_seed(0)
ok = True
for (B, T, cin, cout) in ((3, 999, 512, 512), (2, 1999, 256, 256), (5, 499, 64, 128), (2, 1998, 256, 256)):   # small: with the fp64 check (an even T too: no wrap)
    for dt in (1, 2): ok &= case(B, T, cin, cout, dt, True)
Bb = int(os.environ.get("B", "256"))
for (T, cin, cout) in ((15999, 256, 256), (7999, 256, 256), (3999, 256, 256), (1999, 256, 256), (999, 256, 256)):   # (the shapes are checked against the engine's in DESIGN)
    ok &= case(Bb, T, cin, cout, 1, False)
lib.fh_gemm_set_path(9)
print("ALL BIT-IDENTICAL TO LOOP 7" if ok else "MISMATCH")
