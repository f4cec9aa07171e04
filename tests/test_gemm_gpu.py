"""GPU: the generic MFMA GEMM (fithubert_amd/csrc/gemm.hip) through the C ABI against torch fp32/fp64."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from fithubert_amd import _lib as L  # noqa: E402


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _tdt(dt):
    return torch.float32 if dt == L.DT_F32 else torch.bfloat16


def _tol(dt, K):
    # fp32: exact-fma chains, order differs from the reference -> ~1e-6 * sqrt(K); bf16: inputs are
    # bf16-exact in the test, so only the output rounding (2^-9 relative) + accumulation order remain
    return (2e-5, 1e-5) if dt == L.DT_F32 else (1e-2, 1e-2)


def _run(desc, dt):
    L.check(L.lib().fh_gemm(desc, dt, _stream()), "fh_gemm")
    torch.cuda.synchronize()


@pytest.mark.parametrize("dt", [L.DT_F32, L.DT_BF16])
@pytest.mark.parametrize("xl,yl", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 480, 480), (98, 768, 480), (264, 96, 200)])
def test_gemm_layouts(dt, xl, yl, M, N, K):
    torch.manual_seed(M * 7 + N * 3 + K + xl * 2 + yl)
    dev = "cuda"
    X = torch.randn(M, K, device=dev).to(_tdt(dt))
    Y = torch.randn(N, K, device=dev).to(_tdt(dt))
    ref = (X.double() @ Y.double().t())
    Xs = X if xl == 0 else X.t().contiguous()  # R_STRIDED: stored [K][M]
    Ys = Y if yl == 0 else Y.t().contiguous()
    Cout = torch.full((M, N), float("nan"), device=dev, dtype=_tdt(dt))
    d = L.GemmDesc()
    d.X = L.operand(Xs.data_ptr(), xl, K if xl == 0 else M)
    d.Y = L.operand(Ys.data_ptr(), yl, K if yl == 0 else N)
    d.C = Cout.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    d.ep = L.epilogue()
    if (xl == 1 and M % 8) or (yl == 1 and N % 8):
        pytest.skip("R_STRIDED needs the operand extent to be a multiple of the vector width")
    _run(d, dt)
    rtol, atol = _tol(dt, K)
    err = (Cout.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= atol * scale + rtol * scale, (err, scale)


@pytest.mark.parametrize("dt", [L.DT_F32, L.DT_BF16])
def test_gemm_asymmetric_identity(dt):
    """A = I with an asymmetric B catches a transposed C write (guide section 3)."""
    dev = "cuda"
    M = N = K = 128
    X = torch.eye(M, device=dev).to(_tdt(dt))
    Y = (torch.arange(N, device=dev)[:, None] * 3 + torch.arange(K, device=dev)[None, :] * 0.25).to(_tdt(dt))
    Cout = torch.empty(M, N, device=dev, dtype=_tdt(dt))
    d = L.GemmDesc()
    d.X = L.operand(X.data_ptr(), 0, K)
    d.Y = L.operand(Y.data_ptr(), 0, K)
    d.C = Cout.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    d.ep = L.epilogue()
    _run(d, dt)
    assert torch.equal(Cout.float(), Y.float().t().contiguous())


@pytest.mark.parametrize("dt", [L.DT_F32, L.DT_BF16])
def test_gemm_epilogue_bias_gelu_preact_residual(dt):
    dev = "cuda"
    torch.manual_seed(5)
    M, N, K = 150, 480, 512
    X = torch.randn(M, K, device=dev).to(_tdt(dt))
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(_tdt(dt))
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).to(_tdt(dt))
    Cout = torch.empty(M, N, device=dev, dtype=_tdt(dt))
    pre = torch.empty(M, N, device=dev, dtype=_tdt(dt))
    d = L.GemmDesc()
    d.X = L.operand(X.data_ptr(), 0, K)
    d.Y = L.operand(W.data_ptr(), 0, K)
    d.C = Cout.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    d.ep = L.epilogue(bias=bias.data_ptr(), act=L.ACT_GELU, preact=pre.data_ptr(), residual=res.data_ptr())
    _run(d, dt)
    z = X.double() @ W.double().t() + bias.double()
    ref = torch.nn.functional.gelu(z) + res.double()
    rtol, atol = _tol(dt, K)
    assert (pre.double() - z).abs().max().item() <= (atol + rtol) * z.abs().max().item()
    assert (Cout.double() - ref).abs().max().item() <= (atol + rtol) * ref.abs().max().item() * 2


@pytest.mark.parametrize("dt", [L.DT_F32, L.DT_BF16])
@pytest.mark.parametrize("k,s,cin,cout,T", [(3, 2, 256, 256, 399), (2, 2, 512, 512, 199), (1, 1, 128, 256, 300)])
def test_conv1d_as_gemm_fwd_dgrad_wgrad(dt, k, s, cin, cout, T):
    """Channels-last conv1d = GEMM over overlapping windows; its dgrad (two-segment form for k=3,s=2)
    and wgrad (atomic fp32, index-mapped to the torch [Cout][Cin][k] layout)."""
    dev = "cuda"
    torch.manual_seed(k * 100 + s)
    B = 3
    Tout = (T - k) // s + 1
    x = torch.randn(B, T, cin, device=dev).to(_tdt(dt))
    w = (torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5).to(_tdt(dt))
    wf = w.permute(0, 2, 1).contiguous().view(cout, k * cin)  # [co][j*Cin + ci]
    y = torch.empty(B, Tout, cout, device=dev, dtype=_tdt(dt))
    d = L.GemmDesc()
    d.X = L.operand(x.data_ptr(), 0, s * cin, rpb=Tout, bstride=T * cin)
    d.Y = L.operand(wf.data_ptr(), 0, k * cin)
    d.C = y.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = B * Tout, cout, k * cin, cout, 1, 1
    d.ep = L.epilogue()
    _run(d, dt)
    xr = x.double().permute(0, 2, 1).requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yr = torch.nn.functional.conv1d(xr, wr, stride=s)
    rtol, atol = _tol(dt, k * cin)
    assert (y.double() - yr.permute(0, 2, 1)).abs().max().item() <= (atol + rtol) * yr.abs().max().item()

    dy = torch.randn(B, Tout, cout, device=dev).to(_tdt(dt))
    yr.backward(dy.double().permute(0, 2, 1))
    dx_ref = xr.grad.permute(0, 2, 1)
    dw_ref = wr.grad

    # ---- dgrad
    dx = torch.zeros(B, T, cin, device=dev, dtype=_tdt(dt))  # rows past the last window keep 0
    if k == s or k == 1:
        d = L.GemmDesc()
        d.X = L.operand(dy.data_ptr(), 0, cout)
        d.Y = L.operand(wf.data_ptr(), 1, k * cin)  # Y(j = kk, r = co) = wf[co][kk]
        d.C = dx.data_ptr()
        d.M, d.N, d.Kseg, d.groups, d.splitk = B * Tout, k * cin, cout, 1, 1
        d.ldc, d.c_rpb, d.c_bstride = s * cin, Tout, T * cin
        d.ep = L.epilogue()
        _run(d, dt)
    else:
        assert (k, s) == (3, 2)
        # odd rows: dx[2u+1] = dy[u] @ W_1
        d = L.GemmDesc()
        d.X = L.operand(dy.data_ptr(), 0, cout)
        d.Y = L.operand(wf.data_ptr(), 1, k * cin, segoff=(cin, 0))
        d.C = dx.data_ptr() + cin * dx.element_size()
        d.M, d.N, d.Kseg, d.groups, d.splitk = B * Tout, cin, cout, 1, 1
        d.ldc, d.c_rpb, d.c_bstride = 2 * cin, Tout, T * cin
        d.ep = L.epilogue()
        _run(d, dt)
        # even rows: dx[2u] = dy[u] @ W_0 + dy[u-1] @ W_2,  u = 0..Tout  (two reduction segments)
        n_even = Tout + 1 if 2 * Tout <= T - 1 else Tout
        d = L.GemmDesc()
        d.X = L.operand(dy.data_ptr(), 0, cout, rpb=n_even, bstride=Tout * cout, nseg=2, tshift=(0, -1), tvalid=Tout)
        d.Y = L.operand(wf.data_ptr(), 1, k * cin, nseg=2, segoff=(0, 2 * cin))
        d.C = dx.data_ptr()
        d.M, d.N, d.Kseg, d.groups, d.splitk = B * n_even, cin, cout, 1, 1
        d.ldc, d.c_rpb, d.c_bstride = 2 * cin, n_even, T * cin
        d.ep = L.epilogue()
        _run(d, dt)
    assert (dx.double() - dx_ref).abs().max().item() <= (atol + rtol) * dx_ref.abs().max().item()

    # ---- wgrad: dW[co][ci][j] += sum_m dy[m][co] * xwin[m][j*Cin+ci]   (atomic, split over the reduction)
    dw = torch.zeros(cout, cin, k, device=dev, dtype=torch.float32)
    d = L.GemmDesc()
    d.X = L.operand(dy.data_ptr(), 1, cout)  # X(i = co, r = m)
    d.Y = L.operand(x.data_ptr(), 1, s * cin, rpb=Tout, bstride=T * cin)  # Y(j = kk, r = m) = window(m)[kk]
    d.C = dw.data_ptr()
    d.M, d.N, d.Kseg, d.groups, d.splitk = cout, k * cin, B * Tout, 1, 3
    d.ldc = cin * k
    d.ep = L.epilogue(atomic=1, jdiv=cin, sj1=1, sj0=k)
    _run(d, dt)
    assert (dw.double() - dw_ref).abs().max().item() <= (atol + rtol) * dw_ref.abs().max().item()


@pytest.mark.parametrize("dt", [L.DT_F32, L.DT_BF16])
def test_gemm_groups_dropout_rowmask(dt):
    dev = "cuda"
    torch.manual_seed(9)
    G, B, T, K, N = 3, 2, 50, 480, 480
    M = B * T
    X = torch.randn(G, M, K, device=dev).to(_tdt(dt))
    W = (torch.randn(G, N, K, device=dev) / K ** 0.5).to(_tdt(dt))
    lens = torch.tensor([50, 37], device=dev, dtype=torch.int32)
    out = torch.empty(G, M, N, device=dev, dtype=_tdt(dt))
    d = L.GemmDesc()
    d.X = L.operand(X.data_ptr(), 0, K, gstride=M * K)
    d.Y = L.operand(W.data_ptr(), 0, K, gstride=N * K)
    d.C = out.data_ptr()
    d.M, d.N, d.Kseg, d.groups, d.splitk = M, N, K, G, 1
    d.ldc, d.c_rpb, d.c_bstride, d.c_gstride = N, T, T * N, M * N
    d.ep = L.epilogue(row_len=lens.data_ptr(), drop=(0.25, 1234, 7))
    _run(d, dt)
    ref = torch.einsum("gmk,gnk->gmn", X.double(), W.double()).view(G, B, T, N)
    o = out.double().view(G, B, T, N)
    assert (o[:, 1, 37:] == 0).all()
    kept = o != 0
    frac = kept[:, 0].double().mean().item()
    assert abs(frac - 0.75) < 0.02, frac
    rtol, atol = _tol(dt, K)
    diff = (o - ref / 0.75).abs() * kept
    assert diff.max().item() <= (atol + rtol) * ref.abs().max().item() * 2
    # the mask is a pure function of (seed, site, index): a second launch reproduces it bit for bit
    out2 = torch.empty_like(out)
    d.C = out2.data_ptr()
    _run(d, dt)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("loop", [1, 2, 3, 5, 6, 7, 8])
@pytest.mark.parametrize("opts", ["plain", "bias_gelu", "bias_preact_gelu_drop", "aux_residual", "rowmask_drop",
                                  "gapped_aux_residual", "gapped_bias_preact_gelu_drop_rowmask"])
def test_gemm_epilogue_forms_agree_on_large_tiles(loop, opts):
    """The bf16 epilogue has three forms (bf16-staged in the 8-phase kernel, streamlined and general fp32-staged read-back); which
    one a row block takes depends on the options and on whether the block is full.  M = 600 gives every 256-row tile full 32- and
    128-row blocks plus a ragged tail, N = 480 a partial column tile.  Outputs are checked against fp64 (dropout: the kept set is
    the pure function of (seed, site, index) every form must reproduce -- compared with the register-staging loop 2)."""
    dev = "cuda"
    torch.manual_seed(11)
    M, N, K = 600, 480, 256
    B, T = 3, 200  # row = b * T + t; batch entries contiguous (c_bstride = T * N) or, "gapped", 230 rows apart in memory
    TS = 230 if "gapped" in opts else T
    X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    aux_full = torch.randn(B * TS, N, device=dev).to(torch.bfloat16)
    res_full = torch.randn(B * TS, N, device=dev).to(torch.bfloat16)
    rows = (torch.arange(M, device=dev) // T) * TS + torch.arange(M, device=dev) % T   # memory row of logical row i
    aux, res = aux_full[rows], res_full[rows]
    row_len = torch.tensor([200, 150, 7], dtype=torch.int32, device=dev)

    def run(path):
        L.lib().fh_gemm_set_path(path)
        Cout = torch.full((B * TS, N), float("nan"), device=dev, dtype=torch.bfloat16)
        pre = torch.full((B * TS, N), float("nan"), device=dev, dtype=torch.bfloat16)
        d = L.GemmDesc()
        d.X = L.operand(X.data_ptr(), 0, K)
        d.Y = L.operand(W.data_ptr(), 0, K)
        d.C = Cout.data_ptr()
        d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
        d.c_rpb, d.c_bstride = T, TS * N
        kw = {}
        if "bias" in opts: kw["bias"] = bias.data_ptr()
        if "gelu" in opts: kw["act"] = L.ACT_GELU
        if "preact" in opts: kw["preact"] = pre.data_ptr()
        if "drop" in opts: kw["drop"] = (0.25, 17, 3)
        if "aux" in opts: kw["gelu_aux"] = aux_full.data_ptr()
        if "residual" in opts: kw["residual"] = res_full.data_ptr()
        if "rowmask" in opts: kw["row_len"] = row_len.data_ptr()
        d.ep = L.epilogue(**kw)
        _run(d, L.DT_BF16)
        if TS != T:  # rows between the entries must stay untouched
            gap = torch.ones(B * TS, dtype=torch.bool, device=dev)
            gap[rows] = False
            assert torch.isnan(Cout[gap].float()).all()
        return Cout[rows], pre[rows]

    try:
        got, pre = run(loop)
        base, _ = run(2)
    finally:
        L.lib().fh_gemm_set_path(9)
    z = X.double() @ W.double().t()
    if "bias" in opts: z = z + bias.double()
    ref = torch.nn.functional.gelu(z) if "gelu" in opts else z.clone()
    if "aux" in opts: 
        a = aux.double()
        ref = ref * (0.5 * (1 + torch.erf(a / 2 ** 0.5)) + a * torch.exp(-a * a / 2) / (2 * 3.141592653589793) ** 0.5)
    if "residual" in opts: ref = ref + res.double()
    assert not torch.isnan(got.float()).any()
    scale = ref.abs().max().item()
    if "preact" in opts:
        assert (pre.double() - z).abs().max().item() <= 2e-2 * z.abs().max().item()
    keep = torch.ones(M, N, dtype=torch.bool, device=dev)
    if "drop" in opts:
        keep = base != 0
        assert torch.equal(got != 0, keep)                       # same mask from every epilogue form
        assert abs(keep.double().mean().item() - (0.75 if "rowmask" not in opts else 0.75 * 357 / 600)) < 0.02
        ref = ref / 0.75
    if "rowmask" in opts:
        t = torch.arange(M, device=dev) % T
        b = torch.arange(M, device=dev) // T
        valid = (t < row_len[b].long())[:, None]
        assert torch.equal((got != 0) & ~valid, torch.zeros_like(keep))  # masked rows are exact zeros
        keep = keep & valid
    err = ((got.double() - ref) * keep).abs().max().item()
    assert err <= 2e-2 * scale, (err, scale)


@pytest.mark.parametrize("dt,path", [(L.DT_F32, 9), (L.DT_BF16, 0), (L.DT_BF16, 1), (L.DT_BF16, 2), (L.DT_BF16, 3), (L.DT_BF16, 4),
                                     (L.DT_BF16, 5), (L.DT_BF16, 6), (L.DT_BF16, 7), (L.DT_BF16, 8)])
@pytest.mark.parametrize("wt", [False, True])
@pytest.mark.parametrize("cin,cout,T", [(256, 256, 399), (256, 256, 400), (512, 512, 263), (64, 128, 38)])
def test_conv1d_k3s2_dgrad_as_one_gemm(dt, path, cin, cout, T, wt):
    """The k = 3, s = 2 conv data gradient as ONE product (engine.hip): output row u = frame pair (2u, 2u+1), segment 0 =
    dy[u] against [W_0 | W_1], segment 1 = dy[u-1] against [W_2 | 0] (Y.seg1_limit: never multiplied), times gelu'(aux); with
    an odd number of input frames the last pair's odd frame does not exist (ep.tail_cols) -- the buffer is NaN-poisoned around
    it, the neighbouring batch item's first row must survive.  Every bf16 main loop."""
    dev = "cuda"
    torch.manual_seed(cin + T)
    B, k, s = 3, 3, 2
    Tout = (T - k) // s + 1
    w = (torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5).to(_tdt(dt))
    wf = w.permute(0, 2, 1).contiguous().view(cout, k * cin)
    dy = torch.randn(B, Tout, cout, device=dev).to(_tdt(dt))
    aux = torch.randn(B, T, cin, device=dev).to(_tdt(dt))
    xr = torch.zeros(B, cin, T, device=dev, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv1d(xr, w.double(), stride=s).backward(dy.double().permute(0, 2, 1))
    a = aux.double()
    gp = 0.5 * (1 + torch.erf(a / 2 ** 0.5)) + a * torch.exp(-a * a / 2) / (2 * torch.pi) ** 0.5
    ref = xr.grad.permute(0, 2, 1) * gp
    n_even = Tout + 1 if 2 * Tout <= T - 1 else Tout
    guard = 64
    buf = torch.full((B * T * cin + guard,), float("nan"), device=dev, dtype=_tdt(dt))
    dx = buf[: B * T * cin].view(B, T, cin)
    d = L.GemmDesc()
    d.X = L.operand(dy.data_ptr(), 0, cout, rpb=n_even, bstride=Tout * cout, nseg=2, tshift=(0, -1), tvalid=Tout)
    if wt:   # the engine's bf16 form: weights as [k * Cin][Cout] (reduction index contiguous), segment 1 = the rows of tap 2
        wtt = w.permute(2, 1, 0).contiguous().view(k * cin, cout)
        d.Y = L.operand(wtt.data_ptr(), 0, cout, nseg=2, segoff=(0, 2 * cin * cout), seg1_limit=cin)
    else:
        d.Y = L.operand(wf.data_ptr(), 1, k * cin, nseg=2, segoff=(0, 2 * cin), seg1_limit=cin)
    d.C = dx.data_ptr()
    d.M, d.N, d.Kseg, d.groups, d.splitk = B * n_even, 2 * cin, cout, 1, 1
    d.ldc, d.c_rpb, d.c_bstride = 2 * cin, n_even, T * cin
    d.ep = L.epilogue(gelu_aux=aux.data_ptr(), tail_cols=cin if 2 * n_even > T else 0)
    lib = L.lib()
    try:
        lib.fh_gemm_set_path(path)
        _run(d, dt)
    finally:
        lib.fh_gemm_set_path(9)
    assert torch.isnan(buf[B * T * cin:].float()).all()            # nothing past the buffer
    covered = min(2 * n_even, T)
    assert not torch.isnan(dx[:, :covered].float()).any()
    assert torch.isnan(dx[:, covered:].float()).all()               # rows past the last pair are left to the caller
    rtol, atol = _tol(dt, 2 * cout)
    tol = (atol + rtol) * ref.abs().max().item() * (1.0 if dt == L.DT_F32 else 1.5)   # bf16: the fast gelu' adds <= 1.1e-4
    assert (dx[:, :covered].double() - ref[:, :covered]).abs().max().item() <= tol


@pytest.mark.parametrize("M,N,K,nseg", [(747, 480, 480, 1), (996, 1440, 480, 1), (250, 480, 1440, 1), (130, 96, 520, 1), (747, 480, 960, 1),
                                        (300, 256, 256, 2)])
def test_small_m_loop_one_trip_reduction(M, N, K, nseg):
    """Path 1 (gemm_small_kernel): 64 x 64 tiles whose whole reduction extent comes into LDS in one LDS-DMA burst per 512-deep
    pass -- the loop the per-shape tuner picks for the reference's own batch (3 utterances: 747 rows).  Ragged M / N tiles, K
    that is not a multiple of 32 x 16, several passes (K > 512), a grouped launch and a two-segment reduction with row shifts
    (conv windows), against fp64 and against the register-staging loop 2."""
    dev = "cuda"
    torch.manual_seed(M + N + K)
    G = 3
    X = torch.randn(G, M + 1, K, device=dev).to(torch.bfloat16)   # one spare row: segment 1 reads row t + 1
    W = (torch.randn(G, N, nseg * K, device=dev) / (nseg * K) ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)

    def run(path):
        L.lib().fh_gemm_set_path(path)
        out = torch.full((G, M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        d = L.GemmDesc()
        d.X = L.operand(X.data_ptr(), 0, K, gstride=(M + 1) * K, nseg=nseg, tshift=(0, 1) if nseg == 2 else (0, 0))
        d.Y = L.operand(W.data_ptr(), 0, nseg * K, gstride=N * nseg * K, nseg=nseg, segoff=(0, K) if nseg == 2 else (0, 0))
        d.C = out.data_ptr()
        d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk, d.c_gstride = M, N, K, N, G, 1, M * N
        d.ep = L.epilogue(bias=bias.data_ptr())
        _run(d, L.DT_BF16)
        return out

    try:
        got, base = run(1), run(2)
    finally:
        L.lib().fh_gemm_set_path(9)
    Xd, Wd = X.double(), W.double()
    ref = Xd[:, :M] @ Wd[:, :, :K].transpose(1, 2) + bias.double()
    if nseg == 2:
        ref = ref + Xd[:, 1:M + 1] @ Wd[:, :, K:].transpose(1, 2)
    scale = ref.abs().max().item()
    assert not torch.isnan(got.float()).any()
    assert (got.double() - ref).abs().max().item() <= 2e-2 * scale
    assert (got.double() - base.double()).abs().max().item() <= 1e-2 * scale


@pytest.mark.parametrize("M,N,K,groups,opts", [
    (20480 + 37, 1024, 192, 1, "bias_preact_gelu"),   # 324 tiles, ragged last row tile (its stores are masked and drained), 3 K-tiles
    (70000, 480, 480, 1, "bias"),                      # ragged column tile (a wave whose 64 columns lie past N issues no store), K tail
    (66000, 1440, 64, 1, "bias_gelu"),                 # ONE K-tile per output tile
    (140000, 256, 128, 1, "bias_preact_gelu"),         # two K-tiles: the whole reduction is prefetched under the previous epilogue
    (30000, 512, 256, 3, "bias_gelu_drop"),            # groups flattened into the tile walk; dropout indices
    (66048, 768, 768, 1, "plain"),                     # no bias piece in the prefetch; M a multiple of 256
    (300, 200, 96, 1, "bias_gelu"),                    # fewer tiles than CUs
    (40000, 512, 480, 1, "bias_ystrided"),             # Y read with the output index contiguous (a data gradient's layout)
    (40000, 480, 512, 2, "plain_xstrided_ystrided"),   # both operands with the reduction index as the memory row
])
def test_streaming_8_phase_loop(M, N, K, groups, opts):
    """Loop 8: one workgroup per CU walks tiles f, f + G, ... and issues the next tile's first two K-tiles (and bias row) by LDS-DMA
    before the epilogue of the current one; the waits are COUNTED against the epilogue's own stores (gemm.hip, loop 8).  A wrong
    count reads a tile buffer before its data landed, so every case is compared bit for bit with the one-tile launch (loop 7, same
    arithmetic per tile) and against fp64: many tiles per workgroup, ragged rows / columns, 1 / 2 / many K-tiles, with and without
    the pre-activation copy (16 / 32 stores per wave), grouped launches."""
    dev = "cuda"
    torch.manual_seed(3)
    X = torch.randn(groups, M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(groups, N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(groups, N, device=dev)
    xs, ys = "xstrided" in opts, "ystrided" in opts
    Xm = X.transpose(1, 2).contiguous() if xs else X      # [groups][K][M]: R_STRIDED, leading dimension M
    Wm = W.transpose(1, 2).contiguous() if ys else W      # [groups][K][N]: R_STRIDED, leading dimension N

    def run(path):
        L.lib().fh_gemm_set_path(path)
        out = torch.full((groups, M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        pre = torch.full((groups, M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        d = L.GemmDesc()
        d.X = L.operand(Xm.data_ptr(), 1 if xs else 0, M if xs else K, gstride=M * K)
        d.Y = L.operand(Wm.data_ptr(), 1 if ys else 0, N if ys else K, gstride=N * K)
        d.C = out.data_ptr()
        d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, groups, 1
        d.c_gstride = M * N
        kw = {}
        if "bias" in opts: kw.update(bias=bias.data_ptr(), bias_gstride=N)
        if "gelu" in opts: kw["act"] = L.ACT_GELU
        if "preact" in opts: kw["preact"] = pre.data_ptr()
        if "drop" in opts: kw["drop"] = (0.25, 17, 3)
        d.ep = L.epilogue(**kw)
        for _ in range(3):   # back-to-back launches: a stale tile buffer of the previous launch must not matter either
            _run(d, L.DT_BF16)
        return out, pre

    try:
        got, pre = run(8)
        assert L.lib().fh_gemm_last_loop() == 8
        base, pre7 = run(7)
        if M == 70000:   # the same walk with CUs held back for other streams' kernels (fh_gemm_reserve_cus: data-parallel runs)
            prev = L.lib().fh_gemm_reserve_cus(40)
            try:
                got40, _ = run(8)
            finally:
                L.lib().fh_gemm_reserve_cus(prev)
            assert torch.equal(got40, base)
            # ... and the per-stream form (what a data-parallel run sets on the frozen teacher's stream)
            L.lib().fh_gemm_stream_reserve_cus(_stream(), 40)
            try:
                got40s, _ = run(8)
            finally:
                L.lib().fh_gemm_stream_reserve_cus(_stream(), 0)
            assert torch.equal(got40s, base)
    finally:
        L.lib().fh_gemm_set_path(9)
    assert not torch.isnan(got.float()).any()
    assert torch.equal(got, base)                                  # same arithmetic, tile by tile
    z = torch.bmm(X.double(), W.double().transpose(1, 2))
    if "bias" in opts: z = z + bias.double()[:, None, :]
    if "preact" in opts:
        assert torch.equal(pre, pre7)
        assert (pre.double() - z).abs().max().item() <= 2e-2 * z.abs().max().item()
    ref = torch.nn.functional.gelu(z) if "gelu" in opts else z
    keep = torch.ones_like(got, dtype=torch.bool)
    if "drop" in opts:
        keep = got != 0
        assert abs(keep.double().mean().item() - 0.75) < 0.01
        ref = ref / 0.75
    assert ((got.double() - ref) * keep).abs().max().item() <= 2e-2 * ref.abs().max().item()



@pytest.mark.parametrize("loop", [7, 8])
@pytest.mark.parametrize("K,ys", [(1000, False), (3072 + 24, False), (1000, True)])
def test_8_phase_refill_is_race_free(loop, K, ys):
    """Loops 7 / 8 hand a quarter of a tile buffer back to the LDS-DMA one phase after it was read (K-tile it + 2 lands in the buffer
    K-tile it is being read from; gemm.hip).  A refill that overtook a pending read, or a wait that counted one piece too few, would
    show as a result that changes from launch to launch: 40 launches of a shape with a K tail (zero pieces, the fastest to arrive),
    ragged rows and more tiles than CUs must be bit-identical, and right against fp64."""
    dev = "cuda"
    torch.manual_seed(11)
    M, N = 256 * 300 + 19, 512
    X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    Wm = W.t().contiguous() if ys else W
    bias = torch.randn(N, device=dev)
    outs = [torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16) for _ in range(4)]
    L.lib().fh_gemm_set_path(loop)
    try:
        first = None
        for i in range(40):
            out = outs[i % 4]
            d = L.GemmDesc()
            d.X = L.operand(X.data_ptr(), 0, K)
            d.Y = L.operand(Wm.data_ptr(), 1 if ys else 0, N if ys else K)
            d.C = out.data_ptr()
            d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
            d.ep = L.epilogue(bias=bias.data_ptr())
            _run(d, L.DT_BF16)
            assert L.lib().fh_gemm_last_loop() == loop
            if first is None:
                first = out.clone()
            elif i % 4 == 3:
                for o in outs:
                    assert torch.equal(o, first)
    finally:
        L.lib().fh_gemm_set_path(9)
    ref = X.double() @ W.double().t() + bias.double()
    assert (first.double() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("loop", [8, 7])
@pytest.mark.parametrize("K,ys", [(1000, False), (768, False), (1000, True)])
def test_8_phase_refill_under_load(loop, K, ys):
    """The refill of a tile-buffer quarter that the OTHER wave group may still be reading is ordered by FH_REFILL_GUARD (gemm_loops.cuh:
    this wave's later LDS read has returned => theirs, issued before the barrier, were served).  The window that guard closes moves
    with memory latency, so this runs the loops where the step runs them -- beside another stream: 5,000 launches of a K-tail shape
    (zero pieces: the fastest DMA to arrive) and of a full-K shape on one stream while a second stream keeps the device busy with the
    teacher's conv-layer GEMM shape (loop 7: its workgroups interleave with the launches under test) and a 1 GiB streaming fill
    (co-resident with the persistent loop's workgroups: concurrent HBM / L2 traffic).  Every launch is compared ON THE DEVICE, bit for
    bit, with the result the same loop produced alone on a quiet device, which in turn must equal loop 3's -- the ring loop, whose
    stages are only reused behind a full barrier and which needs no ordering assumption -- to the last bit when the two loops share
    their summation order, within one bf16 rounding otherwise."""
    dev = "cuda"
    torch.manual_seed(21)
    M, N = 256 * 300 + 19, 512
    X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    Wm = W.t().contiguous() if ys else W
    bias = torch.randn(N, device=dev)
    lib = L.lib()

    def desc(out):
        d = L.GemmDesc()
        d.X = L.operand(X.data_ptr(), 0, K)
        d.Y = L.operand(Wm.data_ptr(), 1 if ys else 0, N if ys else K)
        d.C = out.data_ptr()
        d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
        d.ep = L.epilogue(bias=bias.data_ptr())
        return d

    # the second stream's work: a conv-layer-shaped product (k = 3, s = 2 windows of a 512-channel tensor) and a fill
    Tc = 40000
    cx = torch.randn(4, Tc, 512, device=dev).to(torch.bfloat16)
    cw = (torch.randn(512, 1536, device=dev) / 40).to(torch.bfloat16)
    cy = torch.empty(4, (Tc - 3) // 2 + 1, 512, device=dev, dtype=torch.bfloat16)
    fill = torch.empty(1 << 28, device=dev, dtype=torch.float32)
    cd = L.GemmDesc()
    cd.X = L.operand(cx.data_ptr(), 0, 2 * 512, rpb=cy.shape[1], bstride=Tc * 512)
    cd.Y = L.operand(cw.data_ptr(), 0, 1536)
    cd.C = cy.data_ptr()
    cd.M, cd.N, cd.Kseg, cd.ldc, cd.groups, cd.splitk = 4 * cy.shape[1], 512, 1536, 512, 1, 1
    cd.ep = L.epilogue(act=L.ACT_GELU)
    try:
        outs = [torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16) for _ in range(3)]
        lib.fh_gemm_set_path(3)
        _run(desc(outs[0]), L.DT_BF16)
        ref3 = outs[0].clone()
        lib.fh_gemm_set_path(loop)
        _run(desc(outs[0]), L.DT_BF16)
        assert lib.fh_gemm_last_loop() == loop
        quiet = outs[0].clone()
        if not torch.equal(quiet, ref3):   # a different summation order inside a K-tile: then within one rounding of each other
            assert (quiet.double() - ref3.double()).abs().max().item() <= 2.0 ** -7 * ref3.double().abs().max().item()
        bad = torch.zeros((), device=dev, dtype=torch.int64)
        side = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        n_launch = 5000
        for i in range(n_launch):
            if i % 6 == 0:
                with torch.cuda.stream(side):
                    lib.fh_gemm_set_path(7)
                    L.check(lib.fh_gemm(cd, L.DT_BF16, side.cuda_stream), "fh_gemm (side stream)")
                    lib.fh_gemm_set_path(loop)
                    fill.fill_(float(i))
            out = outs[i % 3]
            L.check(lib.fh_gemm(desc(out), L.DT_BF16, main.cuda_stream), "fh_gemm")
            bad += (out != quiet).any()
        main.wait_stream(side)
        torch.cuda.synchronize()
        assert int(bad.item()) == 0, f"{int(bad.item())} of {n_launch} launches differ from the quiet-device result"
    finally:
        lib.fh_gemm_set_path(9)
    ref = X.double() @ W.double().t() + bias.double()
    assert (quiet.double() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("loop", [1, 2, 3, 4, 5, 6, 7, 8])
def test_saved_gelu_derivative_forward_and_backward(loop):
    """fh_gemm_epilogue.aux_factor: the forward stores GELU'(pre-activation) through `preact` (not the pre-activation) and the
    backward multiplies by the stored factor instead of evaluating GELU' -- every epilogue form (M = 600: full and ragged row blocks,
    N = 480: a partial column tile), against fp64."""
    dev = "cuda"
    torch.manual_seed(5)
    M, N, K = 600, 480, 256
    X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    sav = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    d = L.GemmDesc()
    d.X = L.operand(X.data_ptr(), 0, K)
    d.Y = L.operand(W.data_ptr(), 0, K)
    d.C = out.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    d.ep = L.epilogue(bias=bias.data_ptr(), act=L.ACT_GELU, preact=sav.data_ptr(), aux_factor=1)
    dY = torch.randn(M, K, device=dev).to(torch.bfloat16)   # backward product: (dY . W'^T) * saved, W' [N][K] read as R_CONTIG
    dz = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    b = L.GemmDesc()
    b.X = L.operand(dY.data_ptr(), 0, K)
    b.Y = L.operand(W.data_ptr(), 0, K)
    b.C = dz.data_ptr()
    b.M, b.N, b.Kseg, b.ldc, b.groups, b.splitk = M, N, K, N, 1, 1
    b.ep = L.epilogue(gelu_aux=sav.data_ptr(), aux_factor=1)
    try:
        L.lib().fh_gemm_set_path(loop)
        _run(d, L.DT_BF16)
        _run(b, L.DT_BF16)
    finally:
        L.lib().fh_gemm_set_path(9)
    z = X.double() @ W.double().t() + bias.double()
    gp = 0.5 * (1 + torch.erf(z / 2 ** 0.5)) + z * torch.exp(-z * z / 2) / (2 * 3.141592653589793) ** 0.5
    assert (out.double() - torch.nn.functional.gelu(z)).abs().max().item() <= 2e-2 * z.abs().max().item()
    assert (sav.double() - gp).abs().max().item() <= 1e-2            # |GELU'| <= 1.13, bf16 storage
    ref = (dY.double() @ W.double().t()) * sav.double()              # exactly the stored factor
    assert (dz.double() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()


@pytest.mark.parametrize("dt", [L.DT_BF16, L.DT_F16])
@pytest.mark.parametrize("B,T,cin,cout", [(3, 999, 512, 512),    # 1,500 rows: a cut last row tile; every batch entry wraps (odd frame count)
                                          (2, 1999, 256, 256),   # the student's width
                                          (5, 499, 64, 128),     # 128 output columns: the upper two wave columns of a tile lie beyond N
                                          (2, 1998, 256, 256),   # even frame count: no wrap, no tail columns
                                          (300, 257, 64, 64)])   # 129 rows per entry: most 128-row blocks cross an entry boundary
def test_conv_data_gradient_on_the_streaming_loop_with_its_operand(dt, B, T, cin, cout):
    """The k = 3, s = 2 conv data gradient as engine.hip launches it (one product over two shifted reduction segments, output row =
    frame pair, x saved GELU' as a per-element operand, batch entries of an odd frame count wrap and end in a half row): the streaming
    loop's operand instance (gemm_8ph_stream_kernel<.., 1>) against torch's conv1d input gradient in fp64, and bit for bit against the
    non-streaming 8-phase loop in bf16 (both round the product once)."""
    torch.manual_seed(B + T)
    dev, k, s = "cuda", 3, 2
    lib = L.lib()
    tdt = torch.float16 if dt == L.DT_F16 else torch.bfloat16
    Tout = (T - k) // s + 1
    w = (torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5).to(tdt)
    wt = w.permute(2, 1, 0).contiguous().view(k * cin, cout)
    dy = torch.randn(B, Tout, cout, device=dev).to(tdt)
    aux = torch.randn(B, T, cin, device=dev).to(tdt)
    n_even = Tout + 1 if 2 * Tout <= T - 1 else Tout
    outs = {}
    try:
        for path in (7, 8):
            dx = torch.full((B, T, cin), float("nan"), device=dev, dtype=tdt)
            d = L.GemmDesc()
            d.X = L.operand(dy.data_ptr(), 0, cout, rpb=n_even, bstride=Tout * cout, nseg=2, tshift=(0, -1), tvalid=Tout)
            d.Y = L.operand(wt.data_ptr(), 0, cout, nseg=2, segoff=(0, 2 * cin * cout), seg1_limit=cin)
            d.C = dx.data_ptr()
            d.M, d.N, d.Kseg, d.groups, d.splitk = B * n_even, 2 * cin, cout, 1, 1
            d.ldc, d.c_rpb, d.c_bstride = 2 * cin, n_even, T * cin
            d.ep = L.epilogue(gelu_aux=aux.data_ptr(), tail_cols=cin if 2 * n_even > T else 0, aux_factor=1)
            lib.fh_gemm_set_path(path)
            _run(d, dt)
            assert lib.fh_gemm_last_loop() == path
            if 2 * n_even < T:
                dx[:, 2 * n_even:] = 0       # (engine.hip zeroes the frames the conv never reached with fh_zero_rows)
            outs[path] = dx
    finally:
        lib.fh_gemm_set_path(9)
    ref = torch.nn.grad.conv1d_input((B, cin, T), w.double(), dy.double().transpose(1, 2), stride=s).transpose(1, 2) * aux.double()
    assert not torch.isnan(outs[8].float()).any()
    assert (outs[8].double() - ref).abs().max().item() <= {L.DT_BF16: 8e-3, L.DT_F16: 1.2e-3}[dt] * ref.abs().max().item()
    if dt == L.DT_BF16 and T % 2 == 1:   # (with an even frame count loop 7 takes its 16-bit staged epilogue, which rounds twice)
        assert torch.equal(outs[8].view(torch.int16), outs[7].view(torch.int16))


def test_streaming_operand_instance_under_load():
    """The operand instance orders three kinds of LDS-DMA traffic through the two tile buffers with counted vmcnt waits and two extra
    barriers per tile (operand pieces -> K-tile 0 behind the first epilogue half -> K-tile 1 behind the second): 2,000 launches of a conv
    data gradient with many tiles per workgroup beside a second stream that keeps the device busy (a conv-shaped product on loop 7 and a
    streaming fill), every one compared on the device, bit for bit, with the result of the non-streaming 8-phase loop on a quiet device."""
    dev, k, s = "cuda", 3, 2
    torch.manual_seed(5)
    lib = L.lib()
    B, T, cin, cout = 24, 7999, 256, 256           # 96,000 rows = 375 row tiles x 2 column tiles: ~3 tiles per workgroup
    Tout = (T - k) // s + 1
    n_even = Tout + 1 if 2 * Tout <= T - 1 else Tout
    w = (torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5).to(torch.bfloat16)
    wt = w.permute(2, 1, 0).contiguous().view(k * cin, cout)
    dy = torch.randn(B, Tout, cout, device=dev).to(torch.bfloat16)
    aux = torch.randn(B, T, cin, device=dev).to(torch.bfloat16)

    def desc(out):
        d = L.GemmDesc()
        d.X = L.operand(dy.data_ptr(), 0, cout, rpb=n_even, bstride=Tout * cout, nseg=2, tshift=(0, -1), tvalid=Tout)
        d.Y = L.operand(wt.data_ptr(), 0, cout, nseg=2, segoff=(0, 2 * cin * cout), seg1_limit=cin)
        d.C = out.data_ptr()
        d.M, d.N, d.Kseg, d.groups, d.splitk = B * n_even, 2 * cin, cout, 1, 1
        d.ldc, d.c_rpb, d.c_bstride = 2 * cin, n_even, T * cin
        d.ep = L.epilogue(gelu_aux=aux.data_ptr(), tail_cols=cin if 2 * n_even > T else 0, aux_factor=1)
        return d

    Tc = 40000
    cx = torch.randn(4, Tc, 512, device=dev).to(torch.bfloat16)
    cw = (torch.randn(512, 1536, device=dev) / 40).to(torch.bfloat16)
    cy = torch.empty(4, (Tc - 3) // 2 + 1, 512, device=dev, dtype=torch.bfloat16)
    fill = torch.empty(1 << 28, device=dev, dtype=torch.float32)
    cd = L.GemmDesc()
    cd.X = L.operand(cx.data_ptr(), 0, 2 * 512, rpb=cy.shape[1], bstride=Tc * 512)
    cd.Y = L.operand(cw.data_ptr(), 0, 1536)
    cd.C = cy.data_ptr()
    cd.M, cd.N, cd.Kseg, cd.ldc, cd.groups, cd.splitk = 4 * cy.shape[1], 512, 1536, 512, 1, 1
    cd.ep = L.epilogue(act=L.ACT_GELU)
    try:
        outs = [torch.zeros(B, T, cin, device=dev, dtype=torch.bfloat16) for _ in range(3)]
        lib.fh_gemm_set_path(7)
        _run(desc(outs[0]), L.DT_BF16)
        quiet = outs[0].clone()
        lib.fh_gemm_set_path(8)
        _run(desc(outs[1]), L.DT_BF16)
        assert lib.fh_gemm_last_loop() == 8
        assert torch.equal(outs[1], quiet)
        bad = torch.zeros((), device=dev, dtype=torch.int64)
        side = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        n_launch = 2000
        for i in range(n_launch):
            if i % 6 == 0:
                with torch.cuda.stream(side):
                    lib.fh_gemm_set_path(7)
                    L.check(lib.fh_gemm(cd, L.DT_BF16, side.cuda_stream), "fh_gemm (side stream)")
                    lib.fh_gemm_set_path(8)
                    fill.fill_(float(i))
            out = outs[i % 3]
            L.check(lib.fh_gemm(desc(out), L.DT_BF16, main.cuda_stream), "fh_gemm")
            bad += (out != quiet).any()
        main.wait_stream(side)
        torch.cuda.synchronize()
        assert int(bad.item()) == 0, f"{int(bad.item())} of {n_launch} launches differ from the quiet-device result"
    finally:
        lib.fh_gemm_set_path(9)


@pytest.mark.parametrize("path", [7, 8])
@pytest.mark.parametrize("K", [24, 40, 64, 72])
def test_8_phase_loops_with_a_single_or_ragged_k_tile(path, K):
    """Loops 7 / 8 forced onto reductions the tuner never offers them (one K-tile, a K-tile that is mostly tail, one tile + a tail of 8):
    whichever loop launch_path_h ends up on, the result is the fp64 product -- the untuned fallback (gemm.hip launch_16) no longer sends
    such launches there, callers that force a path still may."""
    dev = "cuda"
    torch.manual_seed(K)
    M, N = 40000 + 13, 480
    X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    d = L.GemmDesc()
    d.X, d.Y, d.C = L.operand(X.data_ptr(), 0, K), L.operand(W.data_ptr(), 0, K), out.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    d.ep = L.epilogue(bias=bias.data_ptr())
    L.lib().fh_gemm_set_path(path)
    try:
        for _ in range(3):
            _run(d, L.DT_BF16)
    finally:
        L.lib().fh_gemm_set_path(9)
    ref = X.double() @ W.double().t() + bias.double()
    assert not torch.isnan(out.float()).any()
    assert (out.double() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


def test_untuned_shape_class_takes_the_nearest_measured_one(tmp_path):
    """A launch whose quarter-octave class (tune_key, gemm.hip) was never measured takes the choice of the nearest measured class that
    differs only in the long dimension -- not the 128-tile fallback: M = 64 x 731 rows measured, M = 64 x 768 (the next class) launched
    without tuning; the weight-gradient form (long reduction, atomic) likewise.  Results against fp64."""
    dev = "cuda"
    lib = L.lib()
    saved = str(tmp_path / "table.json")
    L.save_gemm_table(saved)
    lib.fh_gemm_tune(-1)
    torch.manual_seed(0)
    N, K = 480, 480

    def fwd(M, tune, K=K):
        X = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        d = L.GemmDesc()
        d.X, d.Y, d.C = L.operand(X.data_ptr(), 0, K), L.operand(W.data_ptr(), 0, K), out.data_ptr()
        d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
        d.ep = L.epilogue()
        lib.fh_gemm_tune(1 if tune else 0)
        try:
            _run(d, L.DT_BF16)
        finally:
            lib.fh_gemm_tune(0)
        _run(d, L.DT_BF16)
        ref = X.double() @ W.double().t()
        assert (out.double() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
        return lib.fh_gemm_last_loop()

    def wgrad(R, tune):
        A = torch.randn(R, N, device=dev).to(torch.bfloat16)      # dY [R][N]
        Bm = torch.randn(R, K, device=dev).to(torch.bfloat16)     # X  [R][K]
        out = torch.zeros(N, K, device=dev, dtype=torch.float32)
        d = L.GemmDesc()
        d.X, d.Y, d.C = L.operand(A.data_ptr(), 1, N), L.operand(Bm.data_ptr(), 1, K), out.data_ptr()
        d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = N, K, R, K, 1, 8
        d.ep = L.epilogue(atomic=1)
        lib.fh_gemm_tune(1 if tune else 0)
        try:
            _run(d, L.DT_BF16)
        finally:
            lib.fh_gemm_tune(0)
        out.zero_()
        _run(d, L.DT_BF16)
        ref = A.double().t() @ Bm.double()
        assert (out.double() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
        return lib.fh_gemm_last_loop()

    try:
        assert fwd(64 * 768, tune=False) in (7, 8)            # empty table: many rows -> the 256-tile 8-phase loops (streaming: no operand here)
        lib.fh_gemm_tune(-1)
        assert fwd(64 * 768, tune=False, K=64) == 2           # ... but never with a single K-tile: the tuner's own gate (reduction >= 128)
        assert fwd(64 * 768, tune=False, K=40) == 2
        lib.fh_gemm_tune(-1)
        tuned = fwd(64 * 731, tune=True)
        assert tuned != 2
        assert fwd(64 * 768, tune=False) == tuned             # the neighbouring class's measurement
        tw = wgrad(64 * 731, tune=True)
        assert tw != 2
        assert wgrad(64 * 768, tune=False) == tw
        assert wgrad(64 * 300, tune=False) == tw              # an octave away, split count capped by the shorter reduction
    finally:
        lib.fh_gemm_tune(-1)
        L.load_gemm_table(saved)
