"""The `mx` operand form of cvlm_gemm (include/cvlm.h ABI 10): fp16 hi.hi product + the two correction products on the block-scaled
e4m3 matrix instruction.

What is pinned here, operator by operator:
  * the format itself (CPU): hip.mx_pack against its definition, element by element;
  * the PRODUCER: an out_mx launch writes, bit for bit, mx_pack() of the h2 planes the same launch writes without out_mx
    (image bytes, block exponents, lo plane) -- for both LDS-staged epilogue forms that produce mx operands and at a column offset;
  * the CONSUMER: with act = NONE both forms are linear in the accumulator, so (mx launch - split-3 launch) must equal, to fp32
    accumulation noise, the fp64 value of  Whi8.Alo8 + Wlo8.Ahi8 - Whi.Alo - Wlo.Ahi  built from the decoded planes: k-order inside
    the instruction, scale bytes and op_sel of all four unit pairs, ragged last group, K-parts of a partial round, 192-row tiles;
  * end to end accuracy of one launch against the fp64 product of the unquantised operands.
"""
import pytest
import torch
import torch.nn.functional as F

from camouflaged_vlm_amd import hip as H


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32) * scale


def e4m3_values():
    codes = torch.arange(256, dtype=torch.uint8)
    return codes.view(torch.float8_e4m3fn).float()


def test_mx_pack_follows_its_definition():
    """CPU: every byte of mx_pack against the definition in include/cvlm.h, on rows with zeros, massive channels and tiny values."""
    x = rnd(12, 192, seed=1) * 3.0
    x[0, 5] = 2.3e4                     # a massive channel inside a block of O(1) values
    x[1, :32] = 0.0                     # an all-zero block
    x[2, 64:96] *= 1e-6                 # a block of tiny values (fp16 subnormals)
    x[3] *= 1e3
    h = H.H2.pack(x)
    img, sc = H.mx_pack(h)
    R, C = x.shape
    assert img.shape == (R, C // 64, 256) and sc.shape == (R, 4, H.mx_scale_pitch(C))
    hi, lo = h.t[0], h.t[1]
    assert torch.equal(img[:, :, :128].contiguous().view(torch.float16).view(R, C), hi)
    table = e4m3_values()
    for r in range(R):
        for b in range(C // 32):
            blk = hi[r, 32 * b:32 * b + 32].float()
            m = float(blk.abs().max())
            ex = max(int(torch.tensor(m).view(torch.int32) >> 23) & 0xff, 103) - 7
            u, half = b // 2, b % 2
            assert int(sc[r, half, u]) == ex and int(sc[r, 2 + half, u]) == ex - 11
            for src, base, e in ((blk, 128, ex), (lo[r, 32 * b:32 * b + 32].float(), 192, ex - 11)):
                want = src / 2.0 ** (e - 127)
                got = table[img[r, u, base + 32 * half:base + 32 * half + 32].long()]
                assert torch.isfinite(got).all()
                # the nearest e4m3 value (ties aside): never further than half a step of the format at that magnitude
                step = torch.where(want.abs() < 2.0 ** -6, torch.tensor(2.0 ** -9), torch.exp2(torch.floor(torch.log2(want.abs().clamp_min(1e-30))) - 3))
                assert bool(((got - want).abs() <= 0.5 * step + 1e-12).all())
                assert float(want.abs().max()) < 256.0
    # decode helper used by the GPU tests: hi8 / lo8 within 2^-4 (relative to the block's largest value) of hi / lo
    m = H.H2MX.from_planes(h)
    h8, l8 = m.planes8()
    assert float((h8 - hi.float()).abs().max() / hi.float().abs().max()) < 2.0 ** -4


# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    H.load()
    return H


def dev_h2(x):
    return H.H2(H.H2.pack(x).t.cuda())


def to_dev_mx(m: "H.H2MX") -> "H.H2MX":
    return H.H2MX(m.t.cuda(), m.s.cuda(), None if m.lo is None else m.lo.cuda(), m.C, m.c0)


@pytest.mark.gpu
# (4096, 5120, 160): the shape the one-image column split takes (16 x 20 tiles of 256^2 -> one round + the rest as 128^2 tiles): an mx
# image must not take it -- its groups and scale bytes do not move with a plain column offset
@pytest.mark.parametrize("M,N,K,form", [(4096, 1280, 96, "h2res"), (2056, 768, 160, "fold_gelu"), (4096, 5120, 160, "fold_gelu"), (4096, 64, 64, "plain_offset")])
def test_gemm_out_mx_is_mx_pack_of_the_planes(hip, M, N, K, form):
    """Producer side: image, exponents and lo plane of an out_mx launch == mx_pack of the planes the same launch writes as plain h2."""
    from camouflaged_vlm_amd.engine import Linear, LnLinear
    dev = "cuda"
    a = rnd(M, K, seed=3)
    a[:, 7] *= 300.0
    A = dev_h2(a)
    ws = H.new_gemm_workspace(dev)
    if form == "h2res":
        XS = 0.25
        lin = Linear(rnd(N, K, seed=4, scale=0.2), rnd(N, seed=5), dev)
        x_old = rnd(M, N, seed=6)
        x_old[:, 3] *= 1e3
        res = dev_h2(x_old * XS)
        kw = dict(bias=lin.bias, alpha=lin.alpha, residual_h2=(res, 1.0 / XS), out_scale=XS, workspace=ws)
        ref = H.H2.empty(M, N, device=dev)
        st_ref = torch.zeros(H.stats_pieces(N), M, 2, device=dev)
        hip.gemm(A, lin.w, M, N, lin.K, out_h2=ref, row_stats=st_ref, **kw)
        out = H.H2MX.empty(M, N, device=dev, lo_plane=True)
        st = torch.zeros_like(st_ref)
        hip.gemm(A, lin.w, M, N, lin.K, out_h2=out, row_stats=st, **kw)
        assert torch.equal(st, st_ref)
        # ... and read back as an mx residual (hi from the image, lo from the plane) it is the same stream
        out2, ref2 = H.H2.empty(M, N, device=dev), H.H2.empty(M, N, device=dev)
        hip.gemm(A, lin.w, M, N, lin.K, out_h2=ref2, bias=lin.bias, alpha=lin.alpha, residual_h2=(ref, 1.0 / XS), out_scale=XS, workspace=ws)
        hip.gemm(A, lin.w, M, N, lin.K, out_h2=out2, bias=lin.bias, alpha=lin.alpha, residual_h2=(out, 1.0 / XS), out_scale=XS, workspace=ws)
        assert torch.equal(out2.t, ref2.t)
    elif form == "fold_gelu":
        D = K
        gamma, beta = 1.0 + 0.1 * rnd(D, seed=7), 0.05 * rnd(D, seed=8)
        lin = LnLinear(rnd(N, D, seed=9, scale=D ** -0.5), rnd(N, seed=10, scale=0.05), gamma, beta, dev)
        stats = torch.zeros(H.stats_pieces(D), M, 2, device=dev)
        xh = H.H2.empty(M, D, device=dev)
        hip.row_stats_split(a.to(dev), 1.0, xh, stats, M, D)
        merged = torch.zeros(M, 2, device=dev)
        hip.ln_stats_merge(stats, M, D, 1e-6, merged)
        kw = dict(bias=lin.bias, alpha=lin.alpha, act=H.ACT_GELU, out_scale=0.25, ln_fold=(merged, lin.colsum), workspace=ws)
        ref = H.H2.empty(M, N, device=dev)
        hip.gemm(xh, lin.w, M, N, lin.K, out_h2=ref, **kw)
        out = H.H2MX.empty(M, N, device=dev)
        hip.gemm(xh, lin.w, M, N, lin.K, out_h2=out, **kw)
    else:                                                   # the prompt GEMM of the ViT-H blocks: 64 columns at column offset 5120 of a wider operand
        lin = Linear(rnd(N, K, seed=4, scale=0.2), rnd(N, seed=5), dev)
        kw = dict(bias=lin.bias, alpha=lin.alpha, act=H.ACT_GELU, out_scale=0.25, workspace=ws)
        ref = H.H2.empty(M, N, device=dev)
        hip.gemm(A, lin.w, M, N, lin.K, out_h2=ref, **kw)
        wide = H.H2MX.empty(M, 5120 + 64, device=dev)
        wide.t.fill_(0xAB)
        wide.s.fill_(0xCD)
        out = wide.cols(5120)
        hip.gemm(A, lin.w, M, N, lin.K, out_h2=out, **kw)
        assert bool((wide.t[:, :4 * 5120] == 0xAB).all()) and bool((wide.s[:, :, :80] == 0xCD).all()) and bool((wide.s[:, :, 81:] == 0xCD).all())
    torch.cuda.synchronize()
    img, sc = H.mx_pack(H.H2(ref.t.cpu()))
    g0 = out.c0 // 64
    got_img = out.t.cpu().view(M, -1, 256)[:, g0:g0 + N // 64]
    assert torch.equal(got_img[:, :, :128], img[:, :, :128]), "fp16 hi halves"
    assert torch.equal(got_img[:, :, 128:192], img[:, :, 128:192]), "hi8 bytes"
    assert torch.equal(got_img[:, :, 192:], img[:, :, 192:]), "lo8 bytes"
    assert torch.equal(out.s.cpu()[:, :, g0:g0 + N // 64], sc[:, :, :N // 64]), "block exponents"
    if out.lo is not None:
        assert torch.equal(out.lo.cpu(), ref.t[1].cpu()), "lo plane"


def correction_delta(a_mx, w_mx, a_planes, w_planes):
    """fp64: what an mx launch's accumulator holds MORE than a split-3 launch's:  Whi8.Alo8 + Wlo8.Ahi8 - Whi.Alo - Wlo.Ahi."""
    ah8, al8 = (t.double() for t in a_mx.planes8())
    wh8, wl8 = (t.double() for t in w_mx.planes8())
    ah, al = a_planes.t[0].double(), a_planes.t[1].double()
    wh, wl = w_planes.t[0].double(), w_planes.t[1].double()
    return al8 @ wh8.t() + ah8 @ wl8.t() - al @ wh.t() - ah @ wl.t()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,form,why", [
    (2048, 768, 1280, "fold", "8 whole rounds ... 5 groups of 8 units"),
    (4136, 768, 320, "fold", "ragged rows (M % 256 = 40), one whole group + a ragged one (10 units)"),
    (4096, 1280, 5184, "h2res", "lin2 of one image: 80 tiles = a partial round -> K-parts of whole groups; ragged last group (162 units)"),
    (9296, 1024, 4096, "h2res", "CLIP c_proj of the fused forward: 192-row tiles"),
    (2048, 512, 64, "h2res", "a single unit pair"),
])
def test_gemm_mx_consumer_is_the_split3_result_plus_the_e4m3_corrections(hip, M, N, K, form, why):
    dev = "cuda"
    a = rnd(M, K, seed=21)
    a[:, 5::64] *= 16.0                                              # block exponents differ along the row (kept moderate: the h2 OUTPUT
    a[::7, :] *= 0.01                                                #  resolves 2^-22 of a row's largest value, the corrections are 2^-16)
    w = rnd(N, K, seed=22, scale=K ** -0.5)
    w[3::16, :] *= 8.0
    ap, wp = H.H2.pack(a), H.H2.pack(w)
    a_mx, w_mx = H.H2MX.from_planes(ap), H.H2MX.from_planes(wp)
    A_il, W = H.H2IL.from_planes(H.H2(ap.t.to(dev))), H.H2(wp.t.to(dev))
    W_il = H.interleave_planes(W)
    A_mx, W_mx = to_dev_mx(a_mx), to_dev_mx(w_mx)
    ws = H.new_gemm_workspace(dev)
    bias = rnd(N, seed=23).to(dev)
    if form == "fold":
        stats = torch.zeros(H.stats_pieces(K), M, 2, device=dev)
        tmp = H.H2.empty(M, K, device=dev)
        hip.row_stats_split(H.H2(ap.t).float().to(dev), 1.0, tmp, stats, M, K)
        merged = torch.zeros(M, 2, device=dev)
        hip.ln_stats_merge(stats, M, K, 1e-6, merged)
        colsum = (wp.t.double().sum(0).sum(1)).float().to(dev)
        kw = dict(bias=bias, ln_fold=(merged, colsum), workspace=ws)
        factor = merged[:, 0].cpu().double()[:, None]
    else:
        res = dev_h2(rnd(M, N, seed=24))
        kw = dict(bias=bias, residual_h2=(res, 1.0), workspace=ws)
        factor = 1.0
    ref, got = H.H2.empty(M, N, device=dev), H.H2.empty(M, N, device=dev)
    hip.gemm(A_il, W, M, N, K, out_h2=ref, w_il=W_il, **kw)
    hip.gemm(A_mx, W, M, N, K, out_h2=got, w_il=W_il, w_mx=W_mx, **kw)
    torch.cuda.synchronize()
    assert hip.gemm_workspace_errors(ws) == 0
    delta = (got.float().cpu().double() - ref.float().cpu().double())
    want = correction_delta(a_mx, w_mx, ap, wp) * factor
    full = (ap.float().double() @ wp.float().double().t()) * factor
    scale = float(full.abs().max())
    err = float((delta - want).abs().max())
    print(f"mx {form} {M}x{N}x{K} ({why}): |delta - corrections| {err:.2e} of |out| max {scale:.2e}; corrections themselves {float(want.abs().max()):.2e}")
    # fp32 accumulation noise of two kernels that add the same products in different orders is ~ sqrt(K) * 2^-24 of the output; a wrong
    # scale byte, op_sel or k-order would show as an error of the size of the corrections themselves
    assert float(want.abs().max()) > 6 * err, "the check must resolve the corrections"
    assert err < 6e-8 * K ** 0.5 * scale
    # and the launch as a whole against the fp64 product of the unquantised operands: the mode's accuracy on these operands
    exact = (a.double() @ w.double().t()) * factor
    acc = float(((got.float().cpu().double() - ref.float().cpu().double()) + 0.0).abs().max())     # = what mx differs from split 3 by
    print(f"    mx vs split-3 on this launch: {acc:.2e} abs = {acc / scale:.2e} of max |out|")
    assert acc < 2e-4 * scale


@pytest.mark.gpu
def test_mx_pack_on_the_device_equals_the_host(hip):
    """engine.Linear packs its weights on the device (planes, 128-byte-row image, mx image): torch's fp16 / e4m3 conversions there must
    give the bytes of the host's -- tiny values (e4m3 subnormals), zeros and large rows included."""
    w = rnd(512, 1280, seed=5, scale=0.03)
    w[7] *= 300.0
    w[9, :64] = 0.0
    w[11] *= 1e-4
    ph, pd = H.H2.pack(w), H.H2.pack(w.cuda())
    assert torch.equal(ph.t, pd.t.cpu())
    ih, sh = H.mx_pack(ph)
    idv, sdv = H.mx_pack(pd)
    assert torch.equal(ih, idv.cpu()) and torch.equal(sh, sdv.cpu())
    assert torch.equal(H.interleave_planes(ph), H.interleave_planes(pd).cpu())


@pytest.mark.gpu
def test_row_stats_split_mx_writes_mx_pack_of_the_planes(hip):
    """cvlm_row_stats_split_mx (the CLIP stream's seed and its deep-prompt rows): image, exponents, lo plane and piece statistics ==
    those of the planar kernel, through mx_pack -- whole tensor and `copies` rows dropped into a larger stream."""
    dev = "cuda"
    M, D = 581 * 2, 1024
    x = rnd(M, D, seed=31)
    x[:, 9] *= 400.0
    x[5] *= 1e-3
    ref, st_ref = H.H2.empty(M, D, device=dev), torch.zeros(H.stats_pieces(D), M, 2, device=dev)
    hip.row_stats_split(x.to(dev), 0.25, ref, st_ref, M, D)
    out, st = H.H2MX.empty(M, D, device=dev, lo_plane=True), torch.zeros(H.stats_pieces(D), M, 2, device=dev)
    hip.row_stats_split(x.to(dev), 0.25, out, st, M, D)
    torch.cuda.synchronize()
    img, sc = H.mx_pack(H.H2(ref.t.cpu()))
    assert torch.equal(out.t.cpu().view(M, -1, 256), img) and torch.equal(out.s.cpu()[:, :, :D // 64], sc[:, :, :D // 64])
    assert torch.equal(out.lo.cpu(), ref.t[1].cpu()) and torch.equal(st, st_ref)
    # four prompt rows written into rows 577..580 of both images (copies = 2, stride = 581 rows): nothing else moves
    p4 = rnd(4, D, seed=32).to(dev)
    before_t, before_s, before_lo = out.t.clone(), out.s.clone(), out.lo.clone()
    hip.row_stats_split(p4, 0.25, out, st, 4, D, row0=577, copies=2, dst_row_stride=581)
    hip.row_stats_split(p4, 0.25, ref, st_ref, 4, D, row0=577, copies=2, dst_row_stride=581)
    torch.cuda.synchronize()
    img, sc = H.mx_pack(H.H2(ref.t.cpu()))
    assert torch.equal(out.t.cpu().view(M, -1, 256), img) and torch.equal(out.s.cpu()[:, :, :D // 64], sc[:, :, :D // 64])
    assert torch.equal(out.lo.cpu(), ref.t[1].cpu()) and torch.equal(st, st_ref)
    rows = torch.ones(M, dtype=torch.bool)
    rows[577:581] = False
    rows[581 + 577:581 + 581] = False
    assert torch.equal(out.t[rows.to(dev)], before_t[rows.to(dev)]) and torch.equal(out.s[rows.to(dev)], before_s[rows.to(dev)])
    assert torch.equal(out.lo[rows.to(dev)], before_lo[rows.to(dev)])


@pytest.mark.gpu
def test_gemm_mx_argument_checks(hip):
    dev = "cuda"
    M, N, K = 512, 256, 128
    ap, wp = H.H2.pack(rnd(M, K, seed=1)), H.H2.pack(rnd(N, K, seed=2))
    A_mx, W_mx, W = to_dev_mx(H.H2MX.from_planes(ap)), to_dev_mx(H.H2MX.from_planes(wp)), H.H2(wp.t.to(dev))
    out = H.H2.empty(M, N, device=dev)
    with pytest.raises(RuntimeError):                                  # the plain epilogue has no mx instantiation
        hip.gemm(A_mx, W, M, N, K, out_h2=out, w_mx=W_mx)
    with pytest.raises(AssertionError):                                # an mx activation without the weight's mx image
        hip.gemm(A_mx, W, M, N, K, out_h2=out, residual_h2=(out, 1.0))
    with pytest.raises(RuntimeError):                                  # out_mx needs whole 64-column groups
        hip.gemm(H.H2(ap.t.to(dev)), W, M, 200, K, out_h2=H.H2MX.empty(M, 256, device=dev))
